cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_round3.py -x -q -m gpu -k "two_step or fft_32768 or headline or dropped" 2>&1 | tail -8
for opt in "two_step=1" "two_step=0" "two_step=1,two_step_frames=512" "two_step=1,two_step_frames=2048" "two_step=1,two_step_frames=4096"; do
  echo "== offline_batch $opt"; ZEN_HIP_OPTIONS="$opt" python bench.py --workload offline_batch --steps 10 --warmup 3 --no-cpu-baseline | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['x_realtime'], {k: round(v['ms_per_step'],3) for k,v in j['kernels'].items()})"
done
for opt in "two_step=1" "two_step=0"; do
  echo "== offline_long $opt"; ZEN_HIP_OPTIONS="$opt" python bench.py --workload offline_long --steps 10 --warmup 3 --no-cpu-baseline | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['ms_per_step'], j['x_realtime'], {k: round(v['ms_per_step'],3) for k,v in j['kernels'].items()})"
done
for opt in "" "block_fused_minb=6"; do
  echo "== HPR $opt"; ZEN_HIP_OPTIONS="$opt" python bench.py --no-legs --outputs HPR --no-cpu-baseline --no-realtime | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['value'], j['kernel_ms_per_step'])"
done
echo "== P default"; python bench.py --no-legs --no-cpu-baseline --no-realtime | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['value'], j['kernel_ms_per_step'])"
tools/pmc_cmd.sh rt_fused_kernel python3 bench.py --no-legs --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-realtime > gpurun_out/pmc_fused_xcd.json 2>/dev/null
tools/pmc_cmd.sh rt_fused_kernel python3 bench.py --no-legs --steps 3 --warmup 1 --settle-ms 0 --no-cpu-baseline --no-realtime --fused-minb 1 > gpurun_out/pmc_fused_minb1.json 2>/dev/null
python -c "
import json
for f in ('gpurun_out/pmc_fused_xcd.json','gpurun_out/pmc_fused_minb1.json'):
    j=json.load(open(f))
    for k,v in j['kernels'].items(): print(f, k, 'read MB', 2*1024*v.get('FETCH_SIZE',0)/1e6, 'write MB', 1024*v.get('WRITE_SIZE',0)/1e6)
"
