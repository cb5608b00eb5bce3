# full GPU suite + the driver's bench command (what the round-end run does), on one box
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -s --durations=25 -p no:cacheprovider > gpurun_out/r05_gputest2.log 2>&1; echo rc=$? >> gpurun_out/r05_gputest2.log
tail -3 gpurun_out/r05_gputest2.log
python bench.py > gpurun_out/r05_bench2.json 2> gpurun_out/r05_bench2.err; tail -c 600 gpurun_out/r05_bench2.json
