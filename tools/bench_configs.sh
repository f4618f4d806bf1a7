#!/bin/bash
# One bench line per BASELINE.json configuration (per-GPU shard of the multi-GPU ones), appended to $1.
set -o pipefail
OUT=${1:-gpurun_out/configs.jsonl}
: > $OUT
run() { echo "== $*" >&2; python3 bench.py --no-cpu-baseline --no-host-api "$@" 2>/dev/null | tail -n 1 >> $OUT || exit 1; }
run                                                        # configs[1]: bulkscan_null, BXD shape
run --method null-grid                                     # configs[3]: 16-point h2 grid, BXD shape (one GPU's view: full m)
run --method null-grid --m 4445                            # configs[3]: one of 8 shards
run --method alt-grid --steps 5                            # alt-grid, BXD shape
run --n 500 --p 50000 --m 2500 --steps 5 --warmup 2        # configs[2]: one of 8 shards
run --method perms --n 1000 --p 100000 --m 1250 --steps 5 --warmup 2                    # configs[4] shard, fp64
run --method perms --perm-dtype f32 --n 1000 --p 100000 --m 1250 --steps 5 --warmup 2   # configs[4] shard, fp32
