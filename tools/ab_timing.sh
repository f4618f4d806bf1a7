# tools/ab_timing.sh: libraries x {timing marks on, off}
export BLMM_DEV_ENV=1
LIB=bulklmm.jl_amd/csrc/libbulklmm_hip.so
cp $LIB /tmp/lib_orig.so
for r in 1 2 3; do for v in "$@"; do for nt in 0 1; do
  cp $v $LIB
  if [ $nt = 1 ]; then export BLMM_BENCH_NOTIMING=1; else unset BLMM_BENCH_NOTIMING; fi
  python3 bench.py --no-cpu-baseline --no-host-api --no-all-rank-form --steps 200 --warmup 50 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', 'notiming=$nt', round(d['ms_per_step'],4), {k: round(x,4) for k,x in d['phases_ms'].items()}, flush=True)"
done; done; done
cp /tmp/lib_orig.so $LIB
