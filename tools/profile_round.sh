#!/bin/bash
# Evidence run of a round (on the GPU box via gpurun; ROUND=r04 by default): microbenchmarks, the default bench line, kernel-trace stats and the HBM /
# SQ counter passes for every LOD kernel at its BASELINE shape.  Outputs under gpurun_out/$ROUND/; tools/collect_profiles.py
# turns them into the tracked summaries under profiles/.
set -o pipefail
ROUND=${ROUND:-r04}
OUT=gpurun_out/$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== microbenchmarks"
for mb in mb3_f64 mb_f64 mb4_rcp mb_lod; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -I bulklmm.jl_amd/csrc tools/$mb.hip -o /tmp/$mb 2> $OUT/$mb.build.log && /tmp/$mb > $OUT/$mb.log 2>&1 || { echo "$mb failed"; tail -3 $OUT/$mb.build.log $OUT/$mb.log; }
done
echo "== default bench"; python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
echo "== the step without the matrix (blmm_bulkscan_reduced_dev), and socket power with / without the L stores"
python3 bench.py --reduced --no-cpu-baseline > $OUT/bench_reduced.json 2> $OUT/bench_reduced.err || tail -3 $OUT/bench_reduced.err
bash tools/power_probe.sh $OUT/power_null_exact.log || true
bash tools/power_probe.sh $OUT/power_null_exact_reduced.log --reduced || true
prof() {  # tag, kernel regex, bench args...
  local tag=$1 kre=$2; shift 2
  echo "== $tag: kernel trace"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag/trace -o t -- python3 bench.py --no-cpu-baseline --no-host-api --no-all-rank-form --steps 5 --warmup 1 "$@" > $OUT/$tag.trace.log 2>&1 || { tail -5 $OUT/$tag.trace.log; return 1; }
  for ctr in FETCH_SIZE WRITE_SIZE "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE"; do
    local c1=${ctr%% *}
    echo "== $tag: pmc $c1"
    rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex "$kre" --output-format csv -d $OUT/$tag/pmc_$c1 -o p -- python3 bench.py --no-cpu-baseline --no-host-api --no-all-rank-form --steps 3 --warmup 1 "$@" > $OUT/$tag.pmc_$c1.log 2>&1 || { tail -5 $OUT/$tag.pmc_$c1.log; return 1; }
  done
}
prof exact "k_scan_lr|k_scan<0, 2, 4, true, 2, true" || exit 1
prof grid "k_scan<" --method null-grid || exit 1
prof alt "k_scan_alt" --method alt-grid || exit 1
prof perm32 "k_scan_f32" --method perms --perm-dtype f32 --n 1000 --p 100000 --m 1250 || exit 1
# kernel statistics (trace only) of the n >= 500 shards: the own eigensolver, k_rotate_big / k_rotate_f32, k_scan_lr, k_scan_f32
trace_only() {  # tag, bench args...
  local tag=$1; shift
  echo "== $tag: kernel trace"
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$tag/trace -o t -- python3 bench.py --no-cpu-baseline --no-host-api --no-all-rank-form --steps 5 --warmup 1 "$@" > $OUT/$tag.trace.log 2>&1 || { tail -5 $OUT/$tag.trace.log; return 1; }
}
trace_only n500_shard --n 500 --p 50000 --m 2500 || exit 1
trace_only n1000_perm_shard --method perms --n 1000 --p 100000 --m 1250 || exit 1
echo "== one step's timeline"
TRACE_MIN_NS=2000 bash tools/trace_step.sh > $OUT/timeline_bxd_step.txt 2>&1 || true
echo "== every configuration"
bash tools/bench_configs.sh $OUT/configs.jsonl
python3 tools/collect_profiles.py $OUT $ROUND
