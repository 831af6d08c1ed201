# cold runs of the default bench with every track call slower than 0.6 ms reported by the library (its phases)
for i in $(seq 1 ${N:-30}); do ESVIO_FE_SLOW_CALL_MS=0.6 timeout 300 python bench.py --steps 20 --warmup 5 --cpu-frames 0 --no-host-pass --no-sae-pass --no-profile-pass 2>gpurun_out/stall_err.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read()); t=d['tail_latency']; print(d['ms_per_step'], d['repeats']['ms_per_step'], t['per_pass_step_ms_max'])"; grep -h "slow call" gpurun_out/stall_err.txt | cut -c1-400; done
