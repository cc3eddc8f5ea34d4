"""Print the per-kernel summary of a bench.py JSON line (argv[1])."""
import json, sys
d = json.load(open(sys.argv[1]))
print("frames/s", round(d["value"], 1), " ms/step", round(d["ms_per_step"], 1), " single-clip ms", d.get("p50_single_clip_latency_ms"))
r = d.get("roofline", {})
for k, v in r.get("kernels", {}).items():
    print(f"  {k:44s} {v['ms_per_step']:7.1f} ms  {v['avg_launch_us']:7.1f} us  {v['tflops']:6.1f} TF/s  {v['algorithmic_gb_per_s'] or 0:7.1f} GB/s  {v['role']}")
print("  attention ms/step", round(r.get("attention_ms_per_step", 0), 1), " dominant:", r.get("kernel"), " frac", round(r.get("frac", 0), 3))
