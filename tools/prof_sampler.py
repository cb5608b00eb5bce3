"""Sampler leg for profiling: `rocprofv3 --kernel-trace --stats -d OUT -- python3 tools/prof_sampler.py B N [graph|eager] [steps]`
(the program directly after `--`, no wrapper).  Default: HIP-graph replay, 60 reverse steps of the T=1000 schedule."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from view_fusion_amd import sampling_bench, train
model = train.build_model(device="cuda:0", phase="test")
B, N = int(sys.argv[1]), int(sys.argv[2])
graph = (sys.argv[3] if len(sys.argv) > 3 else "graph") == "graph"
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 60
print(sampling_bench.time_sampler(B, N, steps=steps, use_graph=graph, model=model))
