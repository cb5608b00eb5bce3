"""Instruction mix of the MFMA loop of a kernel (fp32 MFMAs do not overlap with VALU on gfx950: count what sits between them).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include --cuda-device-only -S view_fusion_amd/csrc/winograd24.hip -o /tmp/w24.s
    python tools/isa_loop.py /tmp/w24.s wino_conv_kernelILi6ELi0E [--dump]
"""
import collections, re, sys

path, key = sys.argv[1], sys.argv[2]
s = open(path).read()
m = re.search(r"^(_Z\S*%s\S*):.*\n" % re.escape(key), s, re.M)
body = s[m.end():s.index("s_endpgm", m.end())].split("\n")
blocks, cur = [], ["entry", []]
for l in body:
    t = l.strip()
    if not t:
        continue
    if re.match(r"^\.LBB\d+_\d+:", t) or t.startswith("; %bb."):
        blocks.append(cur); cur = [t, []]; continue
    if t.startswith((";", ".")):
        continue
    cur[1].append(t.split(";")[0].strip())
blocks.append(cur)


def cls(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("v_pk_"): return "valu_pk"
    if op.startswith("v_"): return "valu"
    if op in ("s_waitcnt", "s_barrier", "s_nop"): return op
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    return "other"


print(m.group(1))
tot = collections.Counter()
for name, ins in blocks:
    c = collections.Counter(cls(x.split()[0]) for x in ins)
    tot.update(c)
    if len(ins) >= 12:
        print(f"{name[:14]:14s} n={len(ins):4d} " + " ".join(f"{k}={v}" for k, v in sorted(c.items())) + f"   last: {ins[-1][:34]}")
print("total", dict(tot))
loop = [b for b in blocks if sum(1 for x in b[1] if x.startswith("v_mfma")) >= 12]
# the chunk loop may be split over several blocks (exec-masked loads): take all blocks between the first and last MFMA block
idx = [i for i, b in enumerate(blocks) if any(x.startswith("v_mfma") for x in b[1])]
span = blocks[idx[0]:idx[-1] + 1]
c = collections.Counter()
ops = collections.Counter()
for _, ins in span:
    for x in ins:
        c[cls(x.split()[0])] += 1
        if cls(x.split()[0]) in ("valu", "valu_pk"): ops[x.split()[0]] += 1
print("MFMA span (blocks %d..%d):" % (idx[0], idx[-1]), dict(c))
print("  VALU ops:", dict(ops.most_common()))
if "--dump" in sys.argv:
    for name, ins in span:
        print(name)
        for x in ins: print("   ", x)
