#!/bin/bash
# instruction-mix counters of one kernel of a bench.py run: tools/prof_insts.sh <tag> <kernel regex> <bench args...>
set -o pipefail
TAG=$1; KRE=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ctr in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVES GRBM_GUI_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT"; do
  c1=${ctr%% *}
  rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex "$KRE" --output-format csv -d $OUT/$c1 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-api --no-all-rank-form --steps 3 --warmup 1 "$@" > $OUT/$c1.log 2>&1 || { tail -5 $OUT/$c1.log; exit 1; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:50] + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    a = {c: sum(x) / len(x) for c, x in v.items()}
    print(k)
    print("   " + "  ".join(f"{c}={a[c]:.3g}" for c in sorted(a)))
PY
