cd $GRAFT_REPO_ROOT
export ZEN_HIP_OPTIONS="no_persist=1"
tools/pmc_cmd.sh stft_kernel python3 bench.py --workload offline_batch --steps 2 --warmup 1 --settle-ms 0 --no-cpu-baseline > gpurun_out/pmc_ob_stft.json 2>/dev/null
python3 - <<'PY'
import json
j=json.load(open('gpurun_out/pmc_ob_stft.json'))
for k,v in j['kernels'].items():
    print(k)
    for c in sorted(v): print('   %-24s %.4g' % (c, v[c]))
PY
