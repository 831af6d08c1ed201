# round 2, GPU call 21: Arc* ring walk on ranks in registers; grouped compact
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02u
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02u/pytest.log
tail -3 gpurun_out/r02u/pytest.log
timeout 900 python bench.py --cpu-frames 10 --cpu-procs 0 > gpurun_out/r02u/bench_default.json 2> gpurun_out/r02u/bench_default.err
timeout 600 python bench.py --width 1280 --height 720 --rate 1e8 --steps 12 --warmup 3 --repeats 3 --cpu-frames 0 --no-host-pass > gpurun_out/r02u/bench_c5shape.json 2> gpurun_out/r02u/bench_c5shape.err
python - <<'P'
import json
for f in ("bench_default","bench_c5shape"):
    d=json.load(open("gpurun_out/r02u/%s.json"%f))
    print(f, d["value"], d["ms_per_step"], d["repeats"]["ms_per_step"])
    for k,v in d["kernels"].items(): print("  ",k, v["avg_us"], v["launches"], v["achieved_GBs"])
P
