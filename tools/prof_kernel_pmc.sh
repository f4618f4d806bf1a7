#!/bin/bash
# SQ counters of one kernel of a bench.py run: tools/prof_kernel_pmc.sh <tag> <kernel regex> <bench args...>
set -o pipefail
TAG=$1; KRE=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex "$KRE" --output-format csv -d $OUT -o p -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-api --steps 3 --warmup 1 "$@" > $OUT/log.txt 2>&1 || { tail -5 $OUT/log.txt; exit 1; }
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:50] + " grid=" + r.get("Grid_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    a = {c: sum(x) / len(x) for c, x in v.items()}
    cyc = a.get("GRBM_GUI_ACTIVE", 0) / 8
    print(k, "calls", len(next(iter(v.values()))))
    print("   cycles/XCD %.0f  mfma busy %.1f %%  valu inst share %.1f %%  wait_any/wave_cycles %.2f  wait_inst/wave_cycles %.2f" % (
        cyc, 100 * a.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / 1024 / max(cyc, 1), 100 * a.get("SQ_ACTIVE_INST_VALU", 0) / max(a.get("SQ_ACTIVE_INST_ANY", 1), 1),
        a.get("SQ_WAIT_ANY", 0) / max(a.get("SQ_WAVE_CYCLES", 1), 1), a.get("SQ_WAIT_INST_ANY", 0) / max(a.get("SQ_WAVE_CYCLES", 1), 1)))
PY
