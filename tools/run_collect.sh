cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh r03_d > gpurun_out/collect_r03_d.log 2>&1
tail -5 gpurun_out/collect_r03_d.log
cat gpurun_out/prof_r03_d/bench_default.json | head -c 1500
