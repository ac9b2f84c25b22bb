"""Ad-hoc GPU parity sweep (HIP vs oracle) used during bring-up; the pytest suite supersedes it."""
import copy, sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from backends import OraclePSAlign, oracle_api, oracle_swalign, make_pa
from poreseq_amd import synth, _capi
from poreseq_amd.poreseqcpp import PSAlign
from poreseq_amd.util import DEFAULT_PARAMS

P = dict(DEFAULT_PARAMS); P['verbose'] = 0
ok = True
for (L, E, seed, par) in [(300, 5, 1101, P), (240, 4, 1102, dict(P, realign_width=40.0, scoring_width=15.0, point_width=6.0)), (1000, 5, 1001, P)]:
    draft, events, truth = synth.make_region(L, E, seed, oracle_swalign, par)
    hip = _capi.load_hip(); orc = oracle_api()
    for d in (0, 1):
        for e in (0, E - 1):
            ho = []
            for api in (orc, hip):
                h = api.align_create(draft, copy.deepcopy(events), par)
                ho.append(api.debug_fill(h, e, d, events[e].mean.size, len(draft) - 4)); api.align_destroy(h)
            eq = [bool(np.array_equal(x, y, equal_nan=True)) for x, y in zip(*ho)]
            print("L", L, "fill dir", d, "ev", e, eq)
            if not all(eq):
                ok = False
                m0, m1 = ho[0][0], ho[1][0]
                bad = np.argwhere(~((m0 == m1) | (np.isnan(m0) & np.isnan(m1))))
                print("  first mismatches (i,j):", bad[:5].tolist(), [ (m0[tuple(b)], m1[tuple(b)]) for b in bad[:5]])
    t = time.time(); so = make_pa(OraclePSAlign, draft, copy.deepcopy(events), par).ScoreEvents(); to = time.time() - t
    t = time.time(); sh = make_pa(PSAlign, draft, copy.deepcopy(events), par).ScoreEvents(); th = time.time() - t
    print("ScoreEvents equal:", so == sh, "oracle %.3fs hip %.3fs" % (to, th))
    ok &= so == sh
    t = time.time(); po = make_pa(OraclePSAlign, draft, copy.deepcopy(events), par).ScorePoints(); to = time.time() - t
    t = time.time(); ph = make_pa(PSAlign, draft, copy.deepcopy(events), par).ScorePoints(); th = time.time() - t
    same = [a.score == b.score for a, b in zip(po, ph)]
    print("ScorePoints", len(po), "identical:", sum(same), "maxabs", max(abs(a.score - b.score) for a, b in zip(po, ph)), "oracle %.3fs hip %.3fs" % (to, th))
    ok &= all(same)
    if not all(same):
        bad = [i for i, s in enumerate(same) if not s][:8]
        for i in bad: print("   ", po[i].start, repr(po[i].orig), repr(po[i].mut), po[i].score, ph[i].score)
    a = make_pa(OraclePSAlign, draft, copy.deepcopy(events), par); b = make_pa(PSAlign, draft, copy.deepcopy(events), par)
    na, nb = a.Refine(), b.Refine()
    print("Refine", na, nb, a.sequence == b.sequence, all(np.array_equal(x.ref_align, y.ref_align) and np.array_equal(x.ref_like, y.ref_like) for x, y in zip(a.events, b.events)))
    ok &= (na == nb and a.sequence == b.sequence)
print("ALL OK" if ok else "MISMATCH")
