import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import sampling_bench, train
model = train.build_model(device="cuda:0", phase="test")
B, N = int(sys.argv[1]), int(sys.argv[2])
print(sampling_bench.time_sampler(B, N, steps=30, use_graph=False, model=model))
