run() { echo "# bench.py --no-cpu-baseline --no-encoder $*"; timeout 300 python bench.py --no-cpu-baseline --no-encoder --steps 60 --warmup 5 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(json.dumps({k:d[k] for k in ('value','ms_per_step','stages_ms')}), json.dumps({k:d['roofline'][k] for k in ('kernel','frac','launch_ms')}))"; }
run; run --no-surface; run --depth 10; run --depth 10 --no-surface; run --width 7680 --height 4320 --depth 10 --steps 8 --warmup 2; run --width 7680 --height 4320 --depth 10 --steps 8 --warmup 2 --no-surface; run --width 1920 --height 1080 --no-surface
