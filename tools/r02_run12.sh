# round 2, GPU call 9: hist with fused scan (last block done), batched loads
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02l
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -40 > gpurun_out/r02l/pytest.log
tail -3 gpurun_out/r02l/pytest.log
timeout 900 python bench.py --cpu-frames 10 --cpu-procs 0 > gpurun_out/r02l/bench_default.json 2> gpurun_out/r02l/bench_default.err
timeout 600 python bench.py --width 1280 --height 720 --rate 1e8 --steps 12 --warmup 3 --repeats 1 --cpu-frames 0 --no-host-pass > gpurun_out/r02l/bench_c5shape.json 2> gpurun_out/r02l/bench_c5shape.err
python - <<'P'
import json
for f in ("bench_default","bench_c5shape"):
    try:
        d=json.load(open("gpurun_out/r02l/%s.json"%f))
    except Exception as e:
        print(f, "failed", e); continue
    print(f, d["value"], d["ms_per_step"], d.get("repeats"), d.get("host_resident_events"))
    for k,v in d["kernels"].items(): print("  ",k, v["avg_us"], v["launches"], v["achieved_GBs"], (d.get("kernels_replay_schedule") or {}).get(k,{}).get("avg_us"))
P
