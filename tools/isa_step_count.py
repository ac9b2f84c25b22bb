#!/usr/bin/env python3
"""Instruction counts per anti-diagonal of k_fill from the gfx950 ISA (no GPU needed):
    hipcc ... -S --cuda-device-only poreseq_amd/csrc/ps_kernels.hip -o /tmp/k.s ; python3 tools/isa_step_count.py /tmp/k.s [kernel substring]
Splits the kernel at its s_barrier instructions (one per anti-diagonal), classifies every instruction and prints the table of the
steady-state steps (the unrolled 8-step FAST bodies are the runs of similar-sized intervals) plus one step's listing."""
import collections, re, sys

src = open(sys.argv[1]).read().splitlines()
want = sys.argv[2] if len(sys.argv) > 2 else "k_fillILi768ELb1ELb1"
start = next(i for i, l in enumerate(src) if l.startswith("_ZN2ps") and want in l and ": ; @" in l)
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
body = src[start:end]


def cls(op):
    if op.startswith("v_") and ("_f64" in op or "f64" in op): return "valu_f64"
    if op.startswith("v_cndmask") or op.startswith("v_cmp"): return "valu_select"
    if op.startswith("v_"): return "valu_int"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("scratch_") or op.startswith("flat_") or op.startswith("buffer_"): return "vmem"
    if op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"): return "sync"
    if op.startswith("s_"): return "salu"
    return None


ins = []
for l in body:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    c = cls(op)
    if c:
        ins.append((op, c, t))
cuts = [i for i, (op, c, t) in enumerate(ins) if op == "s_barrier"]
steps = [ins[a + 1:b + 1] for a, b in zip(cuts, cuts[1:])]
print("kernel %s: %d instructions, %d barriers" % (want, len(ins), len(cuts)))
rows = []
for k, st in enumerate(steps):
    h = collections.Counter(c for _, c, _ in st)
    rows.append((k, len(st), h))
# steady-state steps: the 8-step bodies; print those with 60 .. 220 instructions
print("%4s %6s | %8s %11s %8s | %5s %5s %5s %5s" % ("step", "total", "valu_f64", "valu_select", "valu_int", "lds", "vmem", "salu", "sync"))
for k, n, h in rows:
    if 60 <= n <= 220:
        print("%4d %6d | %8d %11d %8d | %5d %5d %5d %5d" % (k, n, h["valu_f64"], h["valu_select"], h["valu_int"], h["lds"], h["vmem"], h["salu"], h["sync"]))
if len(sys.argv) > 3:
    k = int(sys.argv[3])
    print("\n---- step %d ----" % k)
    for op, c, t in steps[k]:
        print("   ", t)
