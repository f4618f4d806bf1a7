for r in 1 2; do for w in 5 50 200 800; do python3 bench.py --no-cpu-baseline --no-host-api --no-all-rank-form --steps 20 --warmup $w 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('warmup', $w, round(d['ms_per_step'],4), round(d['phases_ms']['scan'],4), round(d['phases_ms']['eigen'],4))"; done; done
