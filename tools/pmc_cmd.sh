#!/bin/bash
# Hardware counters of one kernel under an arbitrary python command (rocprofv3 --pmc, one pass per counter
# group, FETCH_SIZE and WRITE_SIZE in passes of their own as MI355X_MICROARCH.md prescribes).
# usage: tools/pmc_cmd.sh <kernel-name-substring> python3 <script> [args...]   -> per-dispatch means as JSON
KERN=$1; shift
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
OUT=gpurun_out/pmc_$$; mkdir -p $OUT
i=0
GROUPS_=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE")
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp -d $OUT/p$i --output-format csv -- "$@" > $OUT/p$i.log 2>&1
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
rm -rf $OUT
