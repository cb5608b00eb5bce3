"""Where do torch's fill / copy kernels sit inside one replayed training iteration?  Reads a rocprofv3 kernel trace
(`rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --steps 3 --warmup 2 --no-sampler --no-cpu-baseline --no-roofline`)
and prints, for the last iteration (between the last two adam_multi_kernel launches), every non-library kernel with its neighbours."""
import csv, glob, os, re, sys
path = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("void ", "").replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", n)[:70]
names = [short(r["Kernel_Name"]) for r in rows]
ad = [i for i, n in enumerate(names) if n.startswith("adam_multi_kernel")]
lo, hi = ad[-2] + 1, ad[-1] + 1
seq = names[lo:hi]
print(f"{len(seq)} kernels in the last iteration")
for i, n in enumerate(seq):
    if n.startswith("at::") or "rocclr" in n:
        dur = int(rows[lo + i]["End_Timestamp"]) - int(rows[lo + i]["Start_Timestamp"])
        print(f"{i:4d} {n[:60]:60s} {dur:6d} ns   after: {seq[i - 1][:40] if i else '-':40s} before: {seq[i + 1][:40] if i + 1 < len(seq) else '-'}")
