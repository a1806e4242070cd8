#!/bin/bash
# Hardware counters of the kernels whose name contains <substring>, under an arbitrary command (rocprofv3 --pmc, one
# pass per counter group, FETCH_SIZE and WRITE_SIZE in passes of their own as MI355X_MICROARCH.md prescribes; no
# trace domains beside --pmc).  Put the program itself after the substring (python3 <script> ..., never env / bash -c).
# usage: tools/pmc_cmd.sh <kernel-name-substring> python3 <script> [args...]
#   -> JSON: per-dispatch means PER EXACT KERNEL NAME (template arguments included, namespaces and parameters stripped)
KERN=$1; shift
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
OUT=gpurun_out/pmc_$$; mkdir -p $OUT
i=0
GROUPS_=("SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE")
for grp in "${GROUPS_[@]}"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp -d $OUT/p$i --output-format csv -- "$@" > $OUT/p$i.log 2>&1
done
python3 - "$KERN" $OUT <<'PY'
import csv, glob, json, re, sys, collections
kern, out = sys.argv[1], sys.argv[2]

def short(name):
    """'void ns::(anonymous namespace)::k<true, 0, false>(ns::Args)' -> 'k<true, 0, false>'"""
    name = re.sub(r"^void\s+", "", name.strip())
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):                      # drop the parameter list: the first '(' outside <...> that is
        if ch == "<":                                  # not part of '(anonymous namespace)'
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0 and not name.startswith("(anonymous namespace)", i):
            cut = i
            break
    name = name[:cut]
    return re.sub(r"(?:[A-Za-z_][A-Za-z_0-9]*|\(anonymous namespace\))::", "", name).strip()

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/p*/**/*counter_collection.csv", recursive=True):
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            per[(short(r["Kernel_Name"]), r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
    for (k, d, c), v in per.items():
        acc[k][c].append(v)
res = {}
for k, cs in acc.items():
    res[k] = {c: sum(v) / len(v) for c, v in cs.items()}
    res[k]["dispatches"] = max(len(v) for v in cs.values())
print(json.dumps({"filter": kern, "kernels": res}, indent=1))
PY
rm -rf $OUT
