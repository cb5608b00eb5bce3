#!/bin/bash
# HBM bytes of ONE conv shape's forward kernel (FETCH_SIZE and WRITE_SIZE in separate --pmc passes, gfx950 correction applied):
#   bash tools/hbm_per_shape.sh Cin Cout H      (S = 96; the program after `--` is python3 itself)
set -eu
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
OUT=/tmp/hbm_shape; rm -rf "$OUT"; mkdir -p "$OUT"
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d "$OUT/$c" -- python3 tools/one_wino.py "$1" "$2" "$3" > "$OUT/$c.log" 2>&1 || { echo "pass $c failed"; tail -5 "$OUT/$c.log"; exit 1; }
done
python3 - "$1" "$2" "$3" <<'PY'
import csv, glob, sys
Cin, Cout, H = map(int, sys.argv[1:4])
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for p in glob.glob(f"/tmp/hbm_shape/{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(p)):
            k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
            a = tot.setdefault((k, c), [0.0, 0]); a[0] += float(r["Counter_Value"]); a[1] += 1
alg = 96 * (Cin + Cout) * H * H * 4 / 1e6
for k in sorted({k for k, _ in tot}):
    if "wino" not in k and "conv" not in k: continue
    f = tot.get((k, "FETCH_SIZE"), [0, 1]); w = tot.get((k, "WRITE_SIZE"), [0, 1])
    rd, wr = 2 * f[0] / f[1] * 1024 / 1e6, w[0] / w[1] * 1024 / 1e6
    print(f"{Cin}->{Cout}@{H}x{H} S=96  {k[:40]:40s} launches {f[1]:3d}  read {rd:7.1f} MB  write {wr:7.1f} MB   (x + y = {alg:.1f} MB; x = {96*Cin*H*H*4/1e6:.1f})")
PY
