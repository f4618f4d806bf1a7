#!/bin/bash
# SQ instruction / wait counters of k_backtransform at n = 1000 (two passes; tools/prof_bt_pmc.sh)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/bt_pmc
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM_RD"; do
  tag=$(echo $pass | cut -d" " -f1)
  rocprofv3 --kernel-trace --pmc $pass --kernel-include-regex "k_backtransform" --output-format csv -d $OUT/$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-api --no-all-rank-form --steps 2 --warmup 1 --method perms --perm-dtype f32 --n 1000 --p 20000 --m 256 > $OUT/$tag.log 2>&1 || { tail -3 $OUT/$tag.log; }
done
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, x in sorted(acc.items()):
    print(f"{c:28s} calls {len(x):3d} avg {sum(x)/len(x):16.0f}")
PY
