cd "$(dirname "$0")/../.." || exit 1; mkdir -p gpurun_out
VF_STEP_GRAPH=0 timeout 600 python tools/prof_aten.py > gpurun_out/r05_prof_aten.txt 2>&1
grep -E "^aten::" gpurun_out/r05_prof_aten.txt | head -40
