cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "headline or fused or realtime or hpr_params or blocking or hard_mask_outputs" 2>&1 | tail -3
for opt in "" "rt_fused_diag=7" "" "rt_fused_diag=7"; do
  echo "== P $opt"; ZEN_HIP_OPTIONS="$opt" python bench.py --no-legs --no-cpu-baseline --no-realtime | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['value'], j['kernel_ms_per_step'])"
done
for opt in "" "rt_fused_diag=7"; do
  echo "== HPR $opt"; ZEN_HIP_OPTIONS="$opt" python bench.py --no-legs --outputs HPR --no-cpu-baseline --no-realtime | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['value'], j['kernel_ms_per_step'])"
done
