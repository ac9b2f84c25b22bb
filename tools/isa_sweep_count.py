#!/usr/bin/env python3
"""Instruction counts per step of a strip-sweep kernel from the gfx950 ISA (no GPU needed):
    hipcc <flags of csrc/Makefile> -S --cuda-device-only poreseq_amd/csrc/ps_sweepw.hip -o /tmp/sweepw.s
    python3 tools/isa_sweep_count.py /tmp/sweepw.s k_sweep_wILi4ELi2E [--list]
A multi-wavefront sweep has ONE s_barrier per step inside its main loop (three steps per round for K <= 5): the loop body is the
longest backward-branch span; it is cut at its barriers and every instruction classified.  Prints per-step counts by class and the
per-cell figure (VALU / K)."""
import collections, re, sys

src = open(sys.argv[1]).read().splitlines()
want = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith("_ZN2ps") and want in l and "; @" in l)
end = next(i for i in range(start, len(src)) if "s_endpgm" in src[i])
body = src[start:end]
K = int(re.search(r"ILi(\d+)ELi(\d+)E", want).group(1)) if re.search(r"ILi(\d+)ELi(\d+)E", want) else 1


def cls(op):
    if op.startswith("v_cmp"): return "v_cmp"
    if op.startswith("v_cndmask"): return "v_cndmask"
    if op.startswith("v_") and "f64" in op: return "v_f64"
    if op.startswith("v_mov") or op.startswith("v_accvgpr"): return "v_mov"
    if op.startswith("v_"): return "v_int"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "scratch_", "flat_", "buffer_")): return "vmem"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop", "s_setprio", "s_sleep")): return "sync"
    if op.startswith(("s_load", "s_buffer")): return "smem"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "salu"
    return None

# labels and their line index; find the backward branch spanning the most instructions
labels = {}
ins = []
for l in body:
    t = l.strip()
    m = re.match(r"^(\.L[A-Za-z0-9_]+):", t)
    if m:
        labels[m.group(1)] = len(ins)
        continue
    if not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    c = cls(op)
    if c:
        ins.append((op, c, t))
loops = []
for i, (op, c, t) in enumerate(ins):
    if c == "branch":
        tgt = t.split()[-1]
        if tgt in labels and labels[tgt] < i and i - labels[tgt] > 300:
            loops.append((i - labels[tgt], labels[tgt], i))
# outermost loops only, longest first; "--loop N" picks the N-th (0: the forward sweep of a kept-column kernel, 1: its backward sweep)
loops = sorted([l for l in loops if not any(o is not l and o[1] <= l[1] and l[2] <= o[2] for o in loops)], reverse=True)
which = int(sys.argv[sys.argv.index("--loop") + 1]) if "--loop" in sys.argv else 0
n, a, b = loops[which]
loop = ins[a:b + 1]
nb = sum(1 for op, c, t in loop if op == "s_barrier")
steps = max(nb, 1)
h = collections.Counter(c for _, c, _ in loop)
valu = sum(h[k] for k in ("v_cmp", "v_cndmask", "v_f64", "v_mov", "v_int"))
print("%s: main loop %d instructions, %d barrier(s) => %d step(s) per round, K = %d" % (want, len(loop), nb, steps, K))
for k in ("v_f64", "v_cmp", "v_cndmask", "v_int", "v_mov", "lds", "vmem", "smem", "salu", "branch", "sync"):
    print("  %-10s %6.1f per step  %6.2f per cell" % (k, h[k] / steps, h[k] / steps / K))
print("  %-10s %6.1f per step  %6.2f per cell" % ("VALU", valu / steps, valu / steps / K))
ops = collections.Counter(op for op, c, _ in loop if c.startswith("v_"))
print("  top VALU opcodes per step:", ", ".join("%s %.1f" % (o, n / steps) for o, n in ops.most_common(24)))
if "--list" in sys.argv:
    for op, c, t in loop:
        print("   ", t)
