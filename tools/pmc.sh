#!/bin/bash
# Hardware counters of one kernel during a short default bench run (rocprofv3 --pmc, one pass per group).
# usage: tools/pmc.sh <kernel-name-substring> [bench.py args...]   -> prints per-dispatch means as JSON
# (run on the GPU box:  gpurun -- 'tools/pmc.sh rt_fused_kernel')
KERN=${1:-rt_fused_kernel}; shift
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
OUT=gpurun_out/pmc_$$; mkdir -p $OUT
i=0
GROUPS_=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE")
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $grp -d $OUT/p$i --output-format csv -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-realtime "$@" > $OUT/p$i.log 2>&1
done
python3 - "$KERN" $OUT <<'PY'
import csv, glob, json, sys, collections
kern, out = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (d, c), v in per.items():
        acc[c].append(v)
res = {c: sum(v) / len(v) for c, v in acc.items()}
res["dispatches"] = max((len(v) for v in acc.values()), default=0)
print(json.dumps({"kernel": kern, "per_dispatch_mean": res}, indent=1))
PY
