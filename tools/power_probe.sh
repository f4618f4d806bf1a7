#!/bin/bash
# tools/power_probe.sh OUT [bench args]: samples rocm-smi (socket power, sclk) every ~0.2 s while bench.py loops over many steps --
# is the scan running at a power-managed clock?  
out=$1; shift
python3 bench.py --no-host-api --no-cpu-baseline --no-all-rank-form --steps ${STEPS:-30000} --warmup 5 "$@" > $out.bench.json 2> $out.bench.err &
bp=$!
# the first `import torch` on a fresh box takes a minute or two: sample only once the card is busy
for i in $(seq 1 600); do
  use=$(/opt/rocm/bin/rocm-smi --showuse 2>/dev/null | grep -oE "GPU use \(%\): [0-9]+" | grep -oE "[0-9]+$" | head -1)
  [ "${use:-0}" -ge 50 ] && break
  kill -0 $bp 2>/dev/null || break
  sleep 0.5
done
sleep 2
: > $out
for i in $(seq 1 25); do
  /opt/rocm/bin/rocm-smi --showpower --showclocks --showtemp 2>&1 | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction|hotspot)" >> $out
  echo "--" >> $out
  sleep 0.2
done
wait $bp
/opt/rocm/bin/rocm-smi --showmaxpower 2>&1 | grep -i "max" >> $out
python3 -c "import json; d=json.loads(open('$out.bench.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'], d['phases_ms'])" >> $out
