"""Boundary value types with the reference's shapes (poreseq/Util.py:2-111, Params.py:4-29).

`poreseqcpp.PSAlign` hands these out and accepts them exactly as the reference's
Cython module does (`from Util import MutationInfo, MutationScore`, pyx:13).
"""


class RegionInfo:
    """'name', 'start:end' or 'name:start:end' (poreseq/Util.py:2-33)."""

    def __init__(self, region=None):
        self.start = None
        self.end = None
        self.name = None
        if region is None:
            return
        parts = region.split(':')
        if len(parts) != 2:
            self.name = parts[0]
        if len(parts) > 1:
            self.start = int(parts[-2])
            self.end = int(parts[-1])


def _dot(s):
    return s if len(s) else '.'


class MutationInfo:
    """0-based start, original bases, mutated bases ('' for none) (poreseq/Util.py:35-81)."""

    def __init__(self, info=None):
        self.start = 0
        self.orig = ""
        self.mut = ""
        if info is None:
            return
        if len(info) == 0 or info[0] == '#':
            self.start = -1
            return
        vals = info.split()
        if len(vals) != 3:
            self.start = -1
            return
        self.start = int(vals[0])
        self.orig = '' if vals[1] == '.' else vals[1]
        self.mut = '' if vals[2] == '.' else vals[2]

    def __str__(self):
        return '{}\t{}\t{}'.format(self.start, _dot(self.orig), _dot(self.mut))


class MutationScore:
    """A MutationInfo plus the summed log-likelihood change (poreseq/Util.py:83-111)."""

    def __init__(self):
        self.start = 0
        self.orig = ""
        self.mut = ""
        self.score = 0

    def __str__(self):
        return '{}\t{}\t{}\t{}'.format(self.start, _dot(self.orig), _dot(self.mut), self.score)


# defaults.conf:1-19 of the reference
DEFAULT_PARAMS = {
    'realign_width': 300.0, 'scoring_width': 100.0, 'point_width': 20.0,
    'min_coverage': 0.0, 'max_coverage': 30.0, 'min_overlap': 500.0, 'max_length': 10000.0,
    'end_trim': 150.0, 'lik_offset': 4.5,
    'skip_t': 0.141, 'skip_c': 0.088, 'stay_t': 0.043, 'stay_c': 0.057,
    'extend_t': 0.072, 'extend_c': 0.046, 'insert_t': 0.020, 'insert_c': 0.025,
}


def LoadParams(filename):
    """`key = float` lines; anything else is skipped silently (poreseq/Params.py:4-23)."""
    params = {}
    if filename is None:
        return params
    with open(filename) as f:
        for line in f:
            kv = line.split('=')
            if len(kv) == 2:
                try:
                    params[kv[0].strip()] = float(kv[1])
                except ValueError:
                    pass
    return params


def VaryParams(params, n=16, picks=3, spread=0.15):
    """One training iteration's candidate parameter sets (poreseq/Params.py:31-60): `n` copies of `params`, each with
    `picks` distinct transition parameters (keys ending in _t / _c, in dict order) multiplied by a Gaussian factor
    N(1, spread).  Draws come from Python's global `random`, one `sample` then `picks` `gauss` calls per copy — the
    reference's order, so a seeded generator gives the reference's list."""
    import random
    tunable = [k for k in params.keys() if k.endswith('_t') or k.endswith('_c')]
    out = []
    for _ in range(n):
        cand = params.copy()
        for k in random.sample(tunable, picks):
            cand[k] *= random.gauss(1.0, spread)
        out.append(cand)
    return out


def SaveParams(filename, params):
    with open(filename, 'w') as f:
        for p in params:
            f.write('{} = {}\n'.format(p, params[p]))
