#!/bin/bash
# where the waves of one kernel spend their cycles (SQ counters, one --pmc pass): tools/prof_wait.sh <tag> <kernel regex> <bench args...>
#   SQ_WAIT_ANY = parked at s_waitcnt / barrier; SQ_WAIT_INST_ANY = issue stall; SQ_ACTIVE_INST_ANY = issuing (quad-cycles, disjoint,
#   sum ~ SQ_WAVE_CYCLES); SQ_VALU_MFMA_BUSY_CYCLES in cycles
set -o pipefail
TAG=$1; KRE=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for ctr in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVES" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F64 GRBM_GUI_ACTIVE"; do
  c1=${ctr%% *}
  rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex "$KRE" --output-format csv -d $OUT/$c1 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-api --steps 3 --warmup 1 "$@" > $OUT/$c1.log 2>&1 || { tail -5 $OUT/$c1.log; echo "pass $c1 failed"; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:60] + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(acc.items()):
    a = {c: sum(x) / len(x) for c, x in v.items()}
    print(k)
    print("   " + "  ".join(f"{c}={a[c]:.4g}" for c in sorted(a)))
    if "SQ_WAVE_CYCLES" in a and a["SQ_WAVE_CYCLES"] > 0:
        w = a["SQ_WAVE_CYCLES"]
        print("   share of wave cycles: wait_any %.3f  wait_inst %.3f  active %.3f  (valu %.3f)" % (a.get("SQ_WAIT_ANY", 0) / w, a.get("SQ_WAIT_INST_ANY", 0) / w, a.get("SQ_ACTIVE_INST_ANY", 0) / w, a.get("SQ_ACTIVE_INST_VALU", 0) / w))
PY
