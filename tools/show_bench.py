"""Print the headline and the per-family table of a bench.py JSON line.  usage: show_bench.py FILE"""
import json, sys
d = json.load(open(sys.argv[1]))
print(f"{d['ms_per_step']:.3f} ms/step  {d['value']:.1f} view-steps/s  loss {d['loss']:.7f}")
r = d.get("roofline", {})
if "kernels" in r:
    for k, v in r["kernels"].items():
        print(f"  {k:14s} {v['ms_per_step']:7.3f} ms  n={v['launches_per_step']:4d}  mfma {v.get('frac_of_fp32_mfma_peak', 0):.3f}  hbm {v.get('frac_of_hbm_peak', 0):.3f}")
    print("  unattributed", round(r["unattributed_ms_per_step"], 3), " headline frac", round(r["frac"], 4), " traffic_ratio", r.get("traffic_ratio"))
elif r:
    print(r)
for k, v in d.get("sampler", {}).items():
    if isinstance(v, dict) and "ms_per_step" in v:
        print(f"  sampler {k}: {v['ms_per_step']:.3f} ms/step")
