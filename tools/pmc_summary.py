"""Summarise rocprofv3 --pmc passes into mean counter value per launch per kernel family.

    python tools/pmc_summary.py OUT.json DIR_FETCH DIR_WRITE
Each DIR holds one `rocprofv3 --kernel-trace --pmc <COUNTER> --output-format csv -d DIR -- python bench.py ...`
run (counters are collected in SEPARATE passes, MI355X_MICROARCH.md HBM section).  Values are KiB as
rocprofv3 reports FETCH_SIZE / WRITE_SIZE; the gfx950 x2 correction of FETCH_SIZE is applied by the
reader (bench.py:pmc_traffic), not here.
"""
import csv, glob, json, os, re, sys


def family(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"(attn_fwd_q32_kernel<(?:true|false)>)", name)      # one kernel, two roles: forward | backward (dS + dQ)
    if m:
        return m.group(1)
    m = re.match(r"([\w:]+?)(?:<|\(|$)", name)
    return m.group(1) if m else name


def main(out, *dirs):
    tab = {}
    for d in dirs:
        for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            acc = {}
            for r in csv.DictReader(open(path)):
                k = (family(r["Kernel_Name"]), r["Counter_Name"])
                a = acc.setdefault(k, [0.0, 0])
                a[0] += float(r["Counter_Value"]); a[1] += 1
            for (fam, ctr), (tot, n) in acc.items():
                e = tab.setdefault(fam, {})
                e[ctr] = tot / n
                e["launches"] = n
    # which tree the passes ran on (bench.py prints it next to roofline.traffic_source): VF_PMC_COMMIT, set by the caller
    # -- the GPU box has no .git
    tab["_meta"] = {"commit": os.environ.get("VF_PMC_COMMIT", "unknown"), "command": os.environ.get("VF_PMC_COMMAND", "")}
    json.dump(tab, open(out, "w"), indent=1)
    tab.pop("_meta")
    for fam, e in sorted(tab.items(), key=lambda kv: -(2 * kv[1].get("FETCH_SIZE", 0) + kv[1].get("WRITE_SIZE", 0)) * kv[1]["launches"])[:14]:
        print(f"{fam:40s} launches {e['launches']:5d}  HBM {(2 * e.get('FETCH_SIZE', 0) + e.get('WRITE_SIZE', 0)) / 1024:8.1f} MiB/launch")


if __name__ == "__main__":
    main(*sys.argv[1:])
