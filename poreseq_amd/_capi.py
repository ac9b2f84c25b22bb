"""ctypes binding of the C ABI declared in include/poreseq_hip.h.

`CApi(path)` binds *a* shared library that exports that ABI; the product only ever
binds `poreseq_amd/csrc/libporeseq_hip.so` (see `load_hip`), and raises if it is
missing — there is no CPU fallback.  (The test-suite binds the oracle / reference
shims through the same class, from tests/, never from here.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
HIP_LIB = os.path.join(_HERE, "csrc", "libporeseq_hip.so")

c_dp = C.POINTER(C.c_double)
c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_u8p = C.POINTER(C.c_uint8)


class PsParams(C.Structure):
    # AlignParams, cpp/AlignUtil.h:57-66
    _fields_ = [("lik_offset", C.c_double), ("scoring_width", C.c_int32),
                ("realign_width", C.c_int32), ("verbose", C.c_int32)]


class PoreseqError(Exception):
    pass


# every exported symbol with (restype, argtypes); tests check the library exports all of them
SYMBOLS = {
    "ps_last_error": (C.c_char_p, []),
    "ps_backend_name": (C.c_char_p, []),
    "ps_info": (C.c_int, [C.c_char_p, C.c_int64]),
    "ps_align_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_char_p, C.c_int64, C.c_int32, c_i64p,
                                  c_dp, c_dp, c_dp, c_dp, c_dp, c_dp, C.c_char_p, c_i64p,
                                  C.POINTER(PsParams)]),
    "ps_align_destroy": (None, [C.c_void_p]),
    "ps_align_set_scoring_width": (C.c_int, [C.c_void_p, C.c_int32]),
    "ps_align_new_call": (C.c_int, [C.c_void_p, C.c_int32]),
    "ps_align_n_events": (C.c_int32, [C.c_void_p]),
    "ps_align_n_levels": (C.c_int64, [C.c_void_p, C.c_int32]),
    "ps_align_sequence_length": (C.c_int64, [C.c_void_p]),
    "ps_align_get_sequence": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64]),
    "ps_align_get_event_refs": (C.c_int, [C.c_void_p, C.c_int32, c_dp, c_dp]),
    "ps_muts_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int64, c_i32p, c_i64p, C.c_char_p,
                                 c_i64p, C.c_char_p, c_dp]),
    "ps_muts_destroy": (None, [C.c_void_p]),
    "ps_muts_count": (C.c_int64, [C.c_void_p]),
    "ps_muts_orig_bytes": (C.c_int64, [C.c_void_p]),
    "ps_muts_mut_bytes": (C.c_int64, [C.c_void_p]),
    "ps_muts_export": (C.c_int, [C.c_void_p, c_i32p, c_i64p, C.c_char_p, c_i64p, C.c_char_p, c_dp]),
    "ps_seqs_destroy": (None, [C.c_void_p]),
    "ps_seqs_count": (C.c_int64, [C.c_void_p]),
    "ps_seqs_bytes": (C.c_int64, [C.c_void_p]),
    "ps_seqs_export": (C.c_int, [C.c_void_p, c_i64p, C.c_char_p]),
    "ps_score_alignments": (C.c_int, [C.c_void_p, c_dp, c_dp]),
    "ps_find_point_mutations": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "ps_find_mutations": (C.c_int, [C.c_void_p, C.c_int32, c_i64p, C.c_char_p, C.POINTER(C.c_void_p)]),
    "ps_score_mutations": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "ps_score_mutation_deltas": (C.c_int, [C.c_void_p, C.c_void_p, c_dp]),
    "ps_make_mutations": (C.c_int, [C.c_void_p, C.c_void_p, c_i32p]),
    "ps_viterbi_mutate": (C.c_int, [C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_double,
                                    C.c_double, C.c_int32, C.POINTER(C.c_void_p)]),
    "ps_srand": (C.c_int, [C.c_uint32]),
    "ps_rand_draw": (C.c_int, [C.c_int64, c_dp]),
    "ps_rng_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32]),
    "ps_rng_destroy": (None, [C.c_void_p]),
    "ps_seqs_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int64, c_i64p, C.c_char_p]),
    "ps_batch_score_alignments": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(c_dp), C.POINTER(c_dp)]),
    "ps_batch_find_mutations": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "ps_batch_score_mutations": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]),
    "ps_batch_make_mutations": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), c_i32p]),
    "ps_batch_viterbi_mutate": (C.c_int, [C.c_int32, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int32, C.c_double,
                                          C.c_double, C.c_double, C.c_double, C.POINTER(C.c_void_p)]),
    "ps_swfull": (C.c_int, [C.c_char_p, C.c_int64, C.c_char_p, C.c_int64, c_i32p, c_dp, c_i32p,
                            c_i32p, C.c_int64, c_i64p]),
    "ps_seq_to_states": (C.c_int, [C.c_char_p, C.c_int64, c_i32p, c_i64p]),
    "ps_debug_fill": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, c_dp, c_dp, c_u8p, c_u8p]),
    "ps_set_sweep_min": (C.c_int, [C.c_int32]),
    "ps_set_sweep2_min": (C.c_int, [C.c_int32]),
    "ps_set_sparse_min": (C.c_int, [C.c_int32]),
    "ps_set_sweep_form": (C.c_int, [C.c_int32, C.c_int32]),
    "ps_set_device_fraction": (C.c_int, [C.c_double]),
    "ps_prof_enable": (C.c_int, [C.c_int32]),
    "ps_prof_reset": (C.c_int, []),
    "ps_prof_get": (C.c_int, [C.c_char_p, c_dp, c_i64p, c_dp]),
    "ps_prof_units": (C.c_int, [C.c_char_p, c_dp]),
}


def _dp(a):
    return a.ctypes.data_as(c_dp)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class CApi:
    """One loaded library exporting the include/poreseq_hip.h ABI."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise PoreseqError("poreseq_amd: native library not found: %s "
                               "(build it with `python -c 'import __graft_entry__ as g; g.build()'`)" % path)
        self.path = path
        self.lib = C.CDLL(path)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(self.lib, name)  # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args

    # ------------------------------------------------------------------ helpers
    def check(self, rc):
        if rc != 0:
            msg = self.lib.ps_last_error()
            raise PoreseqError("%s failed (%d): %s" % (os.path.basename(self.path), rc,
                                                      msg.decode() if msg else "?"))

    def backend_name(self):
        return self.lib.ps_backend_name().decode()

    def info(self):
        """process-wide state of the library in one line: stream / hardware-queue mode, runtimes, memory plan"""
        buf = C.create_string_buffer(1024)
        self.check(self.lib.ps_info(buf, 1024))
        return buf.value.decode()

    # ------------------------------------------------------------------ AlignData
    def align_create(self, sequence, events, params):
        """Flatten a PSAlign-shaped object (pyx:139-153) and create the native AlignData."""
        E = len(events)
        for ev in events:
            ev.makecontiguous()
        lens = [int(ev.mean.size) for ev in events]
        off = np.zeros(E + 1, dtype=np.int64)
        off[1:] = np.cumsum(lens)
        cat = lambda name: (_f64(np.concatenate([_f64(getattr(ev, name)) for ev in events]))
                            if E else np.zeros(0))
        mean, stdv, ra, rl = cat("mean"), cat("stdv"), cat("ref_align"), cat("ref_like")
        for ev, n in zip(events, lens):
            if not (ev.stdv.size == n and ev.ref_align.size == n and ev.ref_like.size == n):
                raise PoreseqError("event arrays differ in length")
        model = np.zeros((E, 4, 1024), dtype=np.float64)
        trans = np.zeros((E, 4), dtype=np.float64)
        for e, ev in enumerate(events):
            m = ev.model
            model[e, 0], model[e, 1] = _f64(m.level_mean), _f64(m.level_stdv)
            model[e, 2], model[e, 3] = _f64(m.sd_mean), _f64(m.sd_stdv)
            trans[e] = (m.prob_skip, m.prob_stay, m.prob_extend, m.prob_insert)
        evseqs = [(getattr(ev, "sequence", "") or "").encode("ascii") for ev in events]
        eoff = np.zeros(E + 1, dtype=np.int64)
        eoff[1:] = np.cumsum([len(s) for s in evseqs])
        epool = b"".join(evseqs)
        pp = PsParams(4.5, 150, 300, 0)  # AlignParams defaults, cpp/AlignUtil.h:64
        if "verbose" in params:
            pp.verbose = int(params["verbose"])
        if "lik_offset" in params:
            pp.lik_offset = float(params["lik_offset"])
        if "realign_width" in params:
            pp.realign_width = int(params["realign_width"])  # Python float -> C int truncation (pyx:148-151)
        if "scoring_width" in params:
            pp.scoring_width = int(params["scoring_width"])
        seq = sequence.encode("ascii")
        h = C.c_void_p()
        self.check(self.lib.ps_align_create(C.byref(h), seq, len(seq), E, off.ctypes.data_as(c_i64p),
                                            _dp(mean), _dp(stdv), _dp(ra), _dp(rl), _dp(model),
                                            _dp(trans), epool, eoff.ctypes.data_as(c_i64p), C.byref(pp)))
        return h

    def align_destroy(self, h):
        self.lib.ps_align_destroy(h)

    def align_sequence(self, h):
        n = self.lib.ps_align_sequence_length(h)
        buf = C.create_string_buffer(max(int(n), 1))
        self.check(self.lib.ps_align_get_sequence(h, buf, n))
        return buf.raw[:n].decode("ascii")

    def align_update_events(self, h, events):
        """UpdatePythonEvents (pyx:131-137)."""
        for e, ev in enumerate(events):
            n = int(self.lib.ps_align_n_levels(h, e))
            ra = np.empty(n, dtype=np.float64)
            rl = np.empty(n, dtype=np.float64)
            self.check(self.lib.ps_align_get_event_refs(h, e, _dp(ra), _dp(rl)))
            ev.ref_align[:] = ra
            ev.ref_like[:] = rl

    # ------------------------------------------------------------------ mutation lists
    def muts_create(self, muts, with_scores=False):
        n = len(muts)
        start = np.array([int(m.start) for m in muts], dtype=np.int32)
        origs = [m.orig.encode("ascii") for m in muts]
        mutsb = [m.mut.encode("ascii") for m in muts]
        oo = np.zeros(n + 1, dtype=np.int64)
        oo[1:] = np.cumsum([len(s) for s in origs])
        mo = np.zeros(n + 1, dtype=np.int64)
        mo[1:] = np.cumsum([len(s) for s in mutsb])
        score = np.array([float(m.score) for m in muts], dtype=np.float64) if with_scores else None
        h = C.c_void_p()
        self.check(self.lib.ps_muts_create(C.byref(h), n, start.ctypes.data_as(c_i32p),
                                           oo.ctypes.data_as(c_i64p), b"".join(origs),
                                           mo.ctypes.data_as(c_i64p), b"".join(mutsb),
                                           _dp(score) if score is not None else None))
        return h

    def muts_export(self, h):
        """-> (start int32[n], orig list[str], mut list[str], score float64[n])"""
        n = int(self.lib.ps_muts_count(h))
        ob, mb = int(self.lib.ps_muts_orig_bytes(h)), int(self.lib.ps_muts_mut_bytes(h))
        start = np.zeros(n, dtype=np.int32)
        oo = np.zeros(n + 1, dtype=np.int64)
        mo = np.zeros(n + 1, dtype=np.int64)
        score = np.zeros(n, dtype=np.float64)
        op = C.create_string_buffer(max(ob, 1))
        mp = C.create_string_buffer(max(mb, 1))
        self.check(self.lib.ps_muts_export(h, start.ctypes.data_as(c_i32p), oo.ctypes.data_as(c_i64p), op,
                                           mo.ctypes.data_as(c_i64p), mp, _dp(score)))
        ops, mps = op.raw[:ob].decode("ascii"), mp.raw[:mb].decode("ascii")
        orig = [ops[oo[i]:oo[i + 1]] for i in range(n)]
        mut = [mps[mo[i]:mo[i + 1]] for i in range(n)]
        return start, orig, mut, score

    def muts_destroy(self, h):
        self.lib.ps_muts_destroy(h)

    def seqs_export(self, h):
        n = int(self.lib.ps_seqs_count(h))
        nb = int(self.lib.ps_seqs_bytes(h))
        off = np.zeros(n + 1, dtype=np.int64)
        pool = C.create_string_buffer(max(nb, 1))
        self.check(self.lib.ps_seqs_export(h, off.ctypes.data_as(c_i64p), pool))
        s = pool.raw[:nb].decode("ascii")
        return [s[off[i]:off[i + 1]] for i in range(n)]

    # ------------------------------------------------------------------ free functions
    def score_alignments(self, h, n_events, likes_len=None):
        scores = np.zeros(n_events, dtype=np.float64)
        likes = np.zeros(likes_len, dtype=np.float64) if likes_len is not None else None
        self.check(self.lib.ps_score_alignments(h, _dp(scores), _dp(likes) if likes is not None else None))
        return (scores, likes) if likes is not None else scores

    def find_point_mutations(self, h):
        out = C.c_void_p()
        self.check(self.lib.ps_find_point_mutations(h, C.byref(out)))
        return out

    def find_mutations(self, h, seqs):
        bs = [s.encode("ascii") for s in seqs]
        off = np.zeros(len(bs) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(b) for b in bs])
        out = C.c_void_p()
        self.check(self.lib.ps_find_mutations(h, len(bs), off.ctypes.data_as(c_i64p), b"".join(bs), C.byref(out)))
        return out

    def score_mutations(self, h, hm):
        out = C.c_void_p()
        self.check(self.lib.ps_score_mutations(h, hm, C.byref(out)))
        return out

    def score_mutation_deltas(self, h, hm, n_events, n_muts):
        """[n_events][n_muts] float64: every event's term of every edit's score (their sum in event order + -1e-6 is the score)"""
        out = np.zeros((int(n_events), int(n_muts)), dtype=np.float64)
        if out.size:
            self.check(self.lib.ps_score_mutation_deltas(h, hm, _dp(out)))
        return out

    def make_mutations(self, h, hm):
        nb = C.c_int32(0)
        self.check(self.lib.ps_make_mutations(h, hm, C.byref(nb)))
        return int(nb.value)

    def viterbi_mutate(self, h, nkeep, skip, stay, mmin, mmax, verbose):
        out = C.c_void_p()
        self.check(self.lib.ps_viterbi_mutate(h, nkeep, skip, stay, mmin, mmax, int(bool(verbose)), C.byref(out)))
        try:
            return self.seqs_export(out)
        finally:
            self.lib.ps_seqs_destroy(out)

    # ------------------------------------------------------------------ lock-step batches (several AlignData per call)
    @staticmethod
    def _harr(handles):
        arr = (C.c_void_p * len(handles))()
        for i, h in enumerate(handles):
            arr[i] = h.value if isinstance(h, C.c_void_p) else h
        return arr

    def rng_create(self, seed=1):
        h = C.c_void_p()
        self.check(self.lib.ps_rng_create(C.byref(h), int(seed)))
        return h

    def rng_destroy(self, h):
        self.lib.ps_rng_destroy(h)

    def seqs_create(self, seqs):
        bs = [s.encode("ascii") for s in seqs]
        off = np.zeros(len(bs) + 1, dtype=np.int64)
        off[1:] = np.cumsum([len(b) for b in bs])
        h = C.c_void_p()
        self.check(self.lib.ps_seqs_create(C.byref(h), len(bs), off.ctypes.data_as(c_i64p), b"".join(bs)))
        return h

    def seqs_destroy(self, h):
        self.lib.ps_seqs_destroy(h)

    def batch_score_alignments(self, handles, n_events):
        n = len(handles)
        scores = [np.zeros(max(int(e), 1), dtype=np.float64) for e in n_events]
        sp = (c_dp * n)(*[_dp(a) for a in scores])
        self.check(self.lib.ps_batch_score_alignments(n, self._harr(handles), sp, None))
        return [a[:int(e)] for a, e in zip(scores, n_events)]

    def batch_find_mutations(self, handles, seqs_handles):
        n = len(handles)
        out = (C.c_void_p * n)()
        self.check(self.lib.ps_batch_find_mutations(n, self._harr(handles), self._harr(seqs_handles), out))
        return [C.c_void_p(out[i]) for i in range(n)]

    def batch_score_mutations(self, handles, muts_handles):
        n = len(handles)
        out = (C.c_void_p * n)()
        self.check(self.lib.ps_batch_score_mutations(n, self._harr(handles), self._harr(muts_handles), out))
        return [C.c_void_p(out[i]) for i in range(n)]

    def batch_make_mutations(self, handles, muts_handles):
        n = len(handles)
        nb = np.zeros(max(n, 1), dtype=np.int32)
        self.check(self.lib.ps_batch_make_mutations(n, self._harr(handles), self._harr(muts_handles), nb.ctypes.data_as(c_i32p)))
        return [int(x) for x in nb[:n]]

    def batch_viterbi_mutate(self, handles, rngs, nkeep, skip, stay, mmin, mmax):
        n = len(handles)
        out = (C.c_void_p * n)()
        self.check(self.lib.ps_batch_viterbi_mutate(n, self._harr(handles), self._harr(rngs), nkeep, skip, stay, mmin, mmax, out))
        res = []
        for i in range(n):
            h = C.c_void_p(out[i])
            try:
                res.append(self.seqs_export(h))
            finally:
                self.lib.ps_seqs_destroy(h)
        return res

    def srand(self, seed):
        self.check(self.lib.ps_srand(int(seed)))

    def rand_draw(self, n):
        out = np.zeros(int(n), dtype=np.float64)
        self.check(self.lib.ps_rand_draw(int(n), _dp(out)))
        return out

    def swfull(self, s1, s2):
        b1, b2 = s1.encode("ascii"), s2.encode("ascii")
        cap = len(b1) + len(b2) + 1
        i1 = np.zeros(cap, dtype=np.int32)
        i2 = np.zeros(cap, dtype=np.int32)
        score = C.c_int32(0)
        acc = C.c_double(0)
        n = C.c_int64(0)
        self.check(self.lib.ps_swfull(b1, len(b1), b2, len(b2), C.byref(score), C.byref(acc),
                                      i1.ctypes.data_as(c_i32p), i2.ctypes.data_as(c_i32p), cap, C.byref(n)))
        return int(score.value), float(acc.value), i1[:n.value].copy(), i2[:n.value].copy()

    def seq_to_states(self, s):
        b = s.encode("ascii")
        st = np.zeros(max(len(b), 1), dtype=np.int32)
        n = C.c_int64(0)
        self.check(self.lib.ps_seq_to_states(b, len(b), st.ctypes.data_as(c_i32p), C.byref(n)))
        return st[:n.value].copy()

    def debug_fill(self, h, ev, direction, n_levels, n_states):
        shape = (n_levels + 1, n_states + 1)
        main = np.zeros(shape)
        stay = np.zeros(shape)
        sm = np.zeros(shape, dtype=np.uint8)
        ss = np.zeros(shape, dtype=np.uint8)
        self.check(self.lib.ps_debug_fill(h, ev, direction, _dp(main), _dp(stay),
                                          sm.ctypes.data_as(c_u8p), ss.ctypes.data_as(c_u8p)))
        return main, stay, sm, ss

    def set_sweep_min(self, n):
        """forward-only batches of at least n alignments run one wavefront per alignment (negative: the default)"""
        self.check(self.lib.ps_set_sweep_min(int(n)))

    def set_sweep2_min(self, n):
        """Alignment::update batches (forward + backward sweep per alignment) of at least n sweeps run one wavefront per sweep"""
        self.check(self.lib.ps_set_sweep2_min(int(n)))

    def set_sparse_min(self, n):
        """Alignment::update batches whose edit lists read few matrix columns: strip sweeps with kept columns from n sweeps on"""
        self.check(self.lib.ps_set_sparse_min(int(n)))

    def set_sweep_form(self, rows_per_lane, wavefronts=1):
        """the form strip sweeps try first: rows of the band per lane, wavefronts per (alignment, direction); <= 0: the library's choice"""
        self.check(self.lib.ps_set_sweep_form(int(rows_per_lane), int(wavefronts)))

    def set_device_fraction(self, fraction):
        """the part of the device's memory this process plans for (ranks sharing a GPU: 1 / ranks on it); <= 0: the default"""
        self.check(self.lib.ps_set_device_fraction(float(fraction)))

    def prof_enable(self, on):
        self.check(self.lib.ps_prof_enable(int(on)))   # 1: synchronous per launch, 2: event pairs queued and read by prof_get

    def prof_reset(self):
        self.check(self.lib.ps_prof_reset())

    def prof_units(self, name):
        u = C.c_double(0)
        self.check(self.lib.ps_prof_units(name.encode(), C.byref(u)))
        return float(u.value)

    def prof_get(self, name):
        ms, n, b = C.c_double(0), C.c_int64(0), C.c_double(0)
        self.check(self.lib.ps_prof_get(name.encode(), C.byref(ms), C.byref(n), C.byref(b)))
        return float(ms.value), int(n.value), float(b.value)


_hip = None


def load_hip():
    """The product's one and only native backend.  Fails loudly when it is not built."""
    global _hip
    if _hip is None:
        _hip = CApi(HIP_LIB)
    return _hip
