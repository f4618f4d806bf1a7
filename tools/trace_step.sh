#!/bin/bash
# kernel timeline of one timed step (rocprofv3 --kernel-trace): start/end of every kernel relative to the step's first kernel
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_step
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-host-api --no-cpu-baseline --no-all-rank-form "$@" > $OUT/bench.log 2>&1
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last occurrence of the jacobi kernel starts the last step
idx = [i for i, r in enumerate(rows) if "k_design" in r["Kernel_Name"]]
i0 = idx[-2] if len(idx) > 1 else idx[-1]
t0 = int(rows[i0]["Start_Timestamp"])
i1 = idx[-1] if len(idx) > 1 else len(rows)
for r in rows[i0:i1]:
    if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) < int(__import__("os").environ.get("TRACE_MIN_NS", "9000")): continue      # < 9 us: not shown (TRACE_MIN_NS)
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][:60]
    print(f'{(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} {(int(r["End_Timestamp"]) - t0) / 1e3:9.1f} us  q{r.get("Queue_Id", "?")} {name}')
PY
