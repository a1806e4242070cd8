cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "headline or block_fused or blocking or linearity or multi_stream or golden or finishes or hard_mask_outputs" 2>&1 | tail -2
for opt in "no_direct_out=1" "" "no_direct_out=1" ""; do
  echo "== HPR $opt"; ZEN_HIP_OPTIONS="$opt" python bench.py --no-legs --outputs HPR --no-cpu-baseline --no-realtime | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'], j['kernel_ms_per_step'])"
done
