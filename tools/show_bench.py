"""Print the main figures of a bench.py JSON line:  python tools/show_bench.py <file>"""
import json, sys
d = [json.loads(l) for l in open(sys.argv[1]) if l.startswith("{")][-1]
print("value %.1f kb/s (resident %.1f), %.0f ms per step, load %.0f ms" % (d["value"], d["resident"]["value"], d["ms_per_step"], d["resident"]["load_ms_per_step"]))
print("single region:", d.get("single_region_s"), d.get("single_region_s_all"))
print("north star 1 kb:", json.dumps(d.get("north_star_1kb")))
r = d.get("roofline", {})
print("roofline:", {k: r.get(k) for k in ("kernel", "achieved", "frac", "avg_launch_ms", "traffic", "launches_in_flight_mean", "aggregate_alg_gbs")})
print("fill kernels:", json.dumps(r.get("fill_kernels")))
print("one batch alone:", json.dumps(r.get("one_batch_alone")))
print("cpu:", json.dumps(d.get("cpu_baseline"))[:1800])
print("logl err:", d.get("logl_max_rel_err_vs_cpu"), d.get("accuracy"))
