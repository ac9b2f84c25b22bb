"""Event / model duck types consumed at the boundary.

The native layer reads exactly what the reference's `PythonToEvents` reads
(pyx:99-129): float64 1-D `mean, stdv, ref_align, ref_like`, `sequence`, and
`model.{level_mean, level_stdv, sd_mean, sd_stdv, complement, prob_*}`.  fast5 / BAM
loading (poreseq/EventData.py:100-224, LoadData.py) is out of scope; these classes
only carry arrays and reproduce the two methods the hot path's callers use:
`mapaligns` (EventData.py:226-256) and `setparams` (EventData.py:288-312).
"""
import numpy as np


class PSModel:
    """poreseq/EventData.py:46-78 (same attribute names and fall-back probabilities)."""

    def __init__(self):
        self.level_mean = np.zeros(1024)
        self.level_stdv = np.ones(1024)
        self.sd_mean = np.ones(1024)
        self.sd_stdv = np.ones(1024)
        self.prob_skip = 0.1
        self.prob_stay = 0.1
        self.prob_extend = self.prob_stay
        self.prob_insert = 0.01
        self.name = ''
        self.complement = False


class PSEvent:
    """Array-backed stand-in for poreseq/EventData.py:80-312 (no fast5 access)."""

    def __init__(self, mean, stdv, ref_align=None, ref_like=None, sequence="", model=None):
        self.mean = np.array(mean, dtype=np.float64)
        self.stdv = np.array(stdv, dtype=np.float64)
        n = self.mean.size
        self.ref_align = np.zeros(n) if ref_align is None else np.array(ref_align, dtype=np.float64)
        self.ref_like = np.zeros(n) if ref_like is None else np.array(ref_like, dtype=np.float64)
        self.sequence = sequence
        self.model = model if model is not None else PSModel()
        self.flipped = False

    def makecontiguous(self):
        for obj in (self, self.model):
            for k, v in vars(obj).items():
                if isinstance(v, np.ndarray):
                    setattr(obj, k, np.ascontiguousarray(v, dtype=np.float64))

    def mapaligns(self, pairs):
        """Re-map ref_align through aligned index pairs (EventData.py:226-256): unique in x,
        linear interpolation, round, 0 outside the paired range."""
        pairs = np.asarray(pairs)
        refal = self.ref_align
        keep = refal > 0
        self.ref_align = 0 * self.ref_align
        _, uinds = np.unique(pairs[:, 0], return_index=True)
        pairs = pairs[uinds, :]
        self.ref_align[keep] = np.round(np.interp(refal[keep], pairs[:, 0], pairs[:, 1], 0, 0))
        self.makecontiguous()

    def setparams(self, params):
        """'skip_t' -> model.prob_skip for template models, '_c' for complement (EventData.py:288-312)."""
        for k in params:
            name = 'prob_' + k[:-2]
            if not hasattr(self.model, name):
                continue
            if (k[-2:] == '_t' and not self.model.complement) or (k[-2:] == '_c' and self.model.complement):
                setattr(self.model, name, params[k])
