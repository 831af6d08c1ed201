# round 2, GPU call 15: hist with 4 LDS histograms per barrier pair, wider scan
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02o
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_c5_full_rate_gpu.py tests/test_time_slice_gpu.py -m gpu -x -q -k "sae or c5 or slice or randomized or end_to_end" 2>&1 | tail -3
timeout 900 python bench.py --cpu-frames 0 --repeats 1 --no-host-pass > gpurun_out/r02o/bench_default.json 2> gpurun_out/r02o/bench_default.err
timeout 600 python bench.py --width 1280 --height 720 --rate 1e8 --steps 12 --warmup 3 --repeats 3 --cpu-frames 0 --no-host-pass > gpurun_out/r02o/bench_c5shape.json 2> gpurun_out/r02o/bench_c5shape.err
python - <<'P'
import json
for f in ("bench_default","bench_c5shape"):
    d=json.load(open("gpurun_out/r02o/%s.json"%f))
    print(f, d["value"], d["ms_per_step"], d["repeats"]["ms_per_step"])
    for k,v in d["kernels"].items(): print("  ",k, v["avg_us"], v["launches"], v["achieved_GBs"])
P
