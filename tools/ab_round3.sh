cd $GRAFT_REPO_ROOT
for so in "" "zen_amd/libzen_hip_faketw.so" "" "zen_amd/libzen_hip_faketw.so"; do
  echo "== headline $so"; ZEN_HIP_SO=$so python bench.py --no-cpu-baseline --no-realtime --no-legs --steps 50 | python -c "
import json,sys
j=json.loads(sys.stdin.readline()); print(j['value'], j['ms_per_step'], j['roofline']['avg_launch_ms'])"
done
