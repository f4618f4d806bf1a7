#!/bin/bash
# per-launch durations of the eigensolver's kernels in one bulkscan call: tools/trace_eig.sh N
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_eig
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-host-api --no-cpu-baseline --n ${1:-500} --p 2000 --m 512 > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_design" in r["Kernel_Name"]]
i0 = idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0:]:
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:50]
    if "k_rotate" in name: break
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} +{(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} us  {name}')
PY
