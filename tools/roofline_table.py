"""Per-kernel roofline table from committed evidence: rocprofv3 kernel stats (time) + PMC passes (HBM bytes).

    python tools/roofline_table.py profiles/r01_bench_n1_kernel_stats_v7.csv profiles/r01_bench_pmc_traffic_kib_per_launch.json STEPS
STEPS = step-equivalents the stats run covers (warm-up + timed + instrumented = 15 for the default command).
Achieved HBM rate = (2*FETCH_SIZE + WRITE_SIZE) KiB per launch (gfx950 correction) / mean launch duration.
"""
import csv, json, re, sys

PEAK_BW = 8.0      # TB/s, MI355X_MICROARCH.md
stats, pmc, steps = sys.argv[1], json.load(open(sys.argv[2])), float(sys.argv[3])


def family(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"(attn_fwd_q32_kernel<(?:true|false)>)", name)      # one kernel, two roles: forward | backward (dS + dQ)
    if m:
        return m.group(1)
    m = re.match(r"([\w:]+?)(?:<|\(|$)", name)
    return m.group(1) if m else name


agg = {}
for r in csv.DictReader(open(stats)):
    a = agg.setdefault(family(r["Name"]), [0, 0.0])
    a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"])
tot = sum(v[1] for v in agg.values())
print("| kernel family | launches/step | ms/step | share | avg µs | HBM MB/launch (PMC) | achieved TB/s | of 8 TB/s | bound |")
print("|---|---|---|---|---|---|---|---|---|")
mfma = {"wino_conv_kernel", "wino44_conv_kernel", "wino_wgrad_kernel", "wino44_wgrad_kernel", "conv1x1_bf16x3_kernel", "conv_mfma_kernel", "conv_wgrad_kernel", "attn_fwd_kernel", "attn_fwd_kh_kernel", "attn_fwd_split_kernel", "attn_fwd_q32_kernel<false>", "attn_fwd_q32_kernel<true>", "attn_bwd_dvdk_kernel", "conv1x1_wgrad_kernel", "bgemm_v2_kernel",
        "bgemm_kernel"}
for fam, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if t / tot < 0.002:
        continue
    e = pmc.get(fam)
    avg = t / n * 1e-3
    if e:
        mb = (2 * e.get("FETCH_SIZE", 0.0) + e.get("WRITE_SIZE", 0.0)) * 1024 / 1e6
        bw = mb / avg / 1e6 * 1e6 / 1e6      # MB / us = TB/s
        bws, frac = f"{mb / avg:.2f}", f"{mb / avg / PEAK_BW * 100:.0f} %"
        mbs = f"{mb:.1f}"
    else:
        mbs = bws = frac = "-"
    print(f"| `{fam}` | {n / steps:.1f} | {t / steps / 1e6:.2f} | {t / tot * 100:.1f} % | {avg:.1f} | {mbs} | {bws} | {frac} | "
          f"{'fp32 MFMA' if fam in mfma else 'HBM / latency'} |")
print(f"\ntotal kernel time {tot / steps / 1e6:.2f} ms per step")
