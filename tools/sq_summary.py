"""Mean SQ counter values per launch of one kernel family from rocprofv3 --pmc passes, with the derived figures used in
DESIGN (matrix-pipe busy share, wave-state shares, VALU per MFMA, LDS bank-conflict share).

    rocprofv3 --kernel-trace --pmc <<= 8 SQ counters> --output-format csv -d DIR1 -- python3 tools/one_wino.py Cin Cout H
    rocprofv3 --kernel-trace --pmc <8 more>            --output-format csv -d DIR2 -- python3 tools/one_wino.py Cin Cout H
    python tools/sq_summary.py KERNEL_SUBSTRING DIR1 DIR2
"""
import csv, glob, os, sys

want, dirs = sys.argv[1], sys.argv[2:]
vals, dur = {}, []
for d in dirs:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if want not in r["Kernel_Name"]:
                continue
            a = vals.setdefault(r["Counter_Name"], [0.0, 0])
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if want in r["Kernel_Name"]:
                dur.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) * 1e-3)
m = {k: v[0] / v[1] for k, v in vals.items()}
us = sum(dur) / max(len(dur), 1)
print(f"kernel *{want}*: {len(dur)} launches, mean {us:.1f} us under the counters")
print("  " + "  ".join(f"{k} {v:.0f}" for k, v in sorted(m.items())))
g = m.get
if g("SQ_VALU_MFMA_BUSY_CYCLES") and us:
    # one count per cycle and SIMD with the matrix pipe busy: 1024 SIMDs
    print(f"  matrix pipe busy: {g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024:.0f} cycles per SIMD = "
          f"{g('SQ_VALU_MFMA_BUSY_CYCLES') / 1024 / (us * 2100) * 100:.1f} % of the launch at 2.1 GHz")
if g("SQ_WAVE_CYCLES"):
    w = g("SQ_WAVE_CYCLES")
    print(f"  waves: parked (s_waitcnt / barrier) {100 * g('SQ_WAIT_ANY', 0) / w:.1f} %, stalled at issue "
          f"{100 * g('SQ_WAIT_INST_ANY', 0) / w:.1f} %, issuing {100 * g('SQ_ACTIVE_INST_ANY', 0) / w:.1f} % of SQ_WAVE_CYCLES")
if g("SQ_INSTS_MFMA"):
    print(f"  VALU per MFMA {g('SQ_INSTS_VALU', 0) / g('SQ_INSTS_MFMA'):.2f}, LDS instructions per MFMA {g('SQ_INSTS_LDS', 0) / g('SQ_INSTS_MFMA'):.2f}")
if g("SQ_LDS_IDX_ACTIVE"):
    print(f"  LDS bank-conflict cycles {100 * g('SQ_LDS_BANK_CONFLICT', 0) / g('SQ_LDS_IDX_ACTIVE'):.0f} % of the LDS-active cycles")
