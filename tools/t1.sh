cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/fuzz
for seed in 401 402 403; do
  timeout 200 python tools/fuzz_parity.py --seconds 150 --seed $seed > gpurun_out/fuzz/r03_fuzz_seed$seed.txt 2>&1
  tail -2 gpurun_out/fuzz/r03_fuzz_seed$seed.txt
done
