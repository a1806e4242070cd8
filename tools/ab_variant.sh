for so in "" zen_amd/libzen_hip_preall.so; do
  tag=${so:+PREALL}; tag=${tag:-BASE}
  ZEN_HIP_SO=$so python bench.py --no-cpu-baseline --no-realtime 2>/dev/null | python -c "
import sys,json
l=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=l['config']
print('$tag', 'value', round(l['value']/1e6,2), 'fused_ms', round(l['roofline']['avg_launch_ms'],4), 'all_out', round(c['all_outputs_hops_per_s']/1e6,2), 'sse', round(c['sse_block_hops_per_s']/1e6,2), 'batch', round(c['offline_batch_x_realtime']), 'long', round(c['offline_long_x_realtime']))"
  ZEN_HIP_SO=$so python bench.py --workload offline_batch --steps 10 --warmup 2 --detail 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('$tag batch', round(d['ms_per_step'],4), {a:round(b['ms_per_step'],4) for a,b in k.items()})"
  ZEN_HIP_SO=$so python bench.py --workload offline_long --steps 10 --warmup 2 --detail 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']; print('$tag long', round(d['ms_per_step'],4), {a:round(b['ms_per_step'],4) for a,b in k.items()})"
done
