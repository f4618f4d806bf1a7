#!/bin/bash
# k_lr_panels_w: the span of its workgroups' activity (diagnostic build, s_memrealtime) next to the duration rocprofv3 reports
ROOT=$(pwd)
cd bulklmm.jl_amd/csrc && touch kernels_lowrank.hip && make EXTRA=-DPW_DIAG -j8 > /dev/null 2>&1; cd $ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/pw_prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-host-api --no-cpu-baseline > $OUT/bench.log 2>&1
grep "panels_w diag" $OUT/bench.log | tail -4
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "panels_w" in r["Kernel_Name"]:
        print("rocprof k_lr_panels_w", (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
cd $ROOT/bulklmm.jl_amd/csrc && touch kernels_lowrank.hip && make -j8 > /dev/null 2>&1
