"""Soak check: many regions through one process / thread; device memory must plateau (not a test)."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from poreseq_amd import synth
from poreseq_amd.poreseqcpp import PSAlign, swalign
from poreseq_amd.consensus import consensus_region
from poreseq_amd.util import DEFAULT_PARAMS
P = dict(DEFAULT_PARAMS, verbose=0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
used = []
for k in range(n):
    draft, events, truth = synth.make_region(9000 + 137 * (k % 7), 10, 3000 + k, swalign, P)
    pa = PSAlign(); pa.sequence, pa.events, pa.params = draft, events, dict(P)
    consensus_region(pa, P)
    free, tot = torch.cuda.mem_get_info()
    used.append((tot - free) / 2**30)
    print("region %2d: device memory in use %.2f GiB" % (k, used[-1]), flush=True)
print("growth over the last half: %.2f GiB" % (used[-1] - used[n // 2]))
