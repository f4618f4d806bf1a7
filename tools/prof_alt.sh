#!/bin/bash
# alt-grid kernel: duration (kernel trace) and HBM traffic (FETCH_SIZE / WRITE_SIZE in their own passes) of prebuilt library variants
#   tools/prof_alt.sh ab_libs/lib_x.so ...   -> gpurun_out/prof_alt/<name>.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
LIB=bulklmm.jl_amd/csrc/libbulklmm_hip.so
cp $LIB /tmp/lib_orig.so
OUT=gpurun_out/prof_alt; mkdir -p $OUT
ARGS="--method alt-grid --no-cpu-baseline --no-host-api --steps 3 --warmup 1"
for v in "$@"; do
  tag=$(basename $v .so); cp $v $LIB
  python3 bench.py --method alt-grid --no-cpu-baseline --no-host-api --steps 10 --warmup 2 > $OUT/$tag.bench.json 2> $OUT/$tag.err || { tail -3 $OUT/$tag.err; continue; }
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/$tag.$ctr
    rocprofv3 --kernel-trace --pmc $ctr --kernel-include-regex "k_scan_alt" --output-format csv -d $OUT/$tag.$ctr -o p -- python3 bench.py $ARGS > $OUT/$tag.$ctr.log 2>&1 || tail -3 $OUT/$tag.$ctr.log
  done
  python3 - "$OUT" "$tag" <<'PY'
import csv, glob, json, sys
out, tag = sys.argv[1], sys.argv[2]
d = json.loads(open(f"{out}/{tag}.bench.json").read().strip().splitlines()[-1])
res = {"ms_per_step": d["ms_per_step"], "scan_ms": d["phases_ms"]["scan"]}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = glob.glob(f"{out}/{tag}.{ctr}/**/*counter_collection.csv", recursive=True)
    vals = [float(r["Counter_Value"]) for f in fs for r in csv.DictReader(open(f)) if r["Counter_Name"] == ctr and "k_scan_alt" in r["Kernel_Name"]]
    # the guide's unit: FETCH_SIZE / WRITE_SIZE count in KB on gfx950 ... (tools/collect_profiles.py applies the same corrections)
    res[ctr + "_raw_per_launch"] = sum(vals) / max(len(vals), 1) if vals else None
print(tag, json.dumps(res))
PY
done
cp /tmp/lib_orig.so $LIB
