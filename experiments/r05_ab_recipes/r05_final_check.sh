cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05_smoke_final.log 2>&1; tail -1 gpurun_out/r05_smoke_final.log
bash experiments/r05_ab_recipes/r05_checkpoint.sh
