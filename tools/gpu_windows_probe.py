"""Where do the strip sweeps' DP matrices differ from the oracle's?  (debugging aid; GPU box)
python tools/gpu_windows_probe.py [nw K Wfrom Wto L]"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np, backends as B
from poreseq_amd import synth, _capi
from poreseq_amd.util import DEFAULT_PARAMS
hip, orc = _capi.load_hip(), B.oracle_api()
hip.set_sweep_min(0); hip.set_sweep2_min(0); hip.set_sparse_min(0)
cases = ((1, 4, range(144, 160, 3), 900), (2, 4, range(296, 312, 4), 1500), (4, 2, range(356, 380, 6), 1800), (2, 4, [33], 260))
if len(sys.argv) > 5:
    cases = ((int(sys.argv[1]), int(sys.argv[2]), range(int(sys.argv[3]), int(sys.argv[4])), int(sys.argv[5])),)
for nw, K, widths, L in cases:
    hip.set_sweep_form(K, nw)
    for W in widths:
        P = dict(DEFAULT_PARAMS, verbose=0, realign_width=float(W))
        draft, events, truth = synth.make_region(L, 2, 6000 + W, B.oracle_swalign, P)
        for d in (0, 1):
            out = []
            for api in (hip, orc):
                h = api.align_create(draft, copy.deepcopy(events), P)
                out.append(api.debug_fill(h, 1, d, events[1].mean.size, len(draft) - 4))
                api.align_destroy(h)
            for k in range(4 if d == 0 else 2):
                a, b = out[0][k], out[1][k]
                if a.dtype.kind == "f":
                    bad = ~((a == b) | (np.isnan(a) & np.isnan(b)))
                else:
                    bad = a != b
                if bad.any():
                    ii, jj = np.where(bad)
                    first = [(int(i), int(j), a[i, j].item(), b[i, j].item()) for i, j in list(zip(ii, jj))[:6]]
                    # band rows of the first bad column (from the oracle's main matrix)
                    j0 = int(jj.min())
                    rows = np.where(~np.isnan(out[1][0][1:, j0]))[0] + 1
                    prow = np.where(~np.isnan(out[1][0][1:, j0 - 1]))[0] + 1 if j0 > 1 else rows
                    print("DIFF nw=%d K=%d W=%d dir=%d k=%d: %d cells; first column %d band [%d, %d] prev [%d, %d]; %s" % (
                        nw, K, W, d, k, int(bad.sum()), j0, rows.min(), rows.max(), prow.min(), prow.max(), first), flush=True)
        print("done nw=%d K=%d W=%d" % (nw, K, W), flush=True)
