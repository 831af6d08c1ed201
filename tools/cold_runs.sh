# N cold runs of the default bench (headline passes only): pass 0, the repeats, every pass's slowest step
for i in $(seq 1 ${N:-24}); do timeout 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-host-pass --no-sae-pass --no-profile-pass 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['tail_latency']; print(d['ms_per_step'], d['value'], d['repeats']['ms_per_step'], 'max step of each pass', t['per_pass_step_ms_max'])"; done
