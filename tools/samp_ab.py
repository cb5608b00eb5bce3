import os, sys
sys.path.insert(0, os.getcwd())
from view_fusion_amd import sampling_bench, train
model = train.build_model(device="cuda:0", phase="test")
out = []
for B, N in ((1, 1), (1, 6), (1, 12), (2, 6)):
    r = sampling_bench.time_sampler(B, N, steps=100, model=model, use_graph=None)
    out.append(round(r["ms_per_step"], 3))
print("KSPLIT_WGS=" + os.environ.get("VF_CONV_KSPLIT_WGS", "256"), out)
