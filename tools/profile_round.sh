#!/bin/bash
# Runs on the GPU box (via gpurun): default bench, rocprofv3 kernel-trace stats and the HBM PMC passes for the
# dominant kernel.  Outputs under gpurun_out/<tag>/; copy the summaries into profiles/ afterwards.
set -o pipefail
TAG=${1:-r01}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== bench (default flags)"; python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
cat $OUT/bench.json
echo "== kernel trace + stats"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
echo "== pmc FETCH_SIZE"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex "k_scan" --output-format csv -d $OUT/pmc_fetch -o fetch -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
echo "== pmc WRITE_SIZE"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex "k_scan" --output-format csv -d $OUT/pmc_write -o write -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_write.log 2>&1 || { tail -5 $OUT/pmc_write.log; exit 1; }
echo "== pmc SQ"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex "k_scan" --output-format csv -d $OUT/pmc_sq -o sq -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OUT/pmc_sq.log 2>&1 || { tail -5 $OUT/pmc_sq.log; exit 1; }
ls -R $OUT | head -40
