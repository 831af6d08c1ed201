# round 2, GPU call 13: benches + rocprofv3 passes (profiles r02d)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ad
timeout 900 python bench.py > gpurun_out/r02ad/bench_default.json 2> gpurun_out/r02ad/bench_default.err
timeout 600 python bench.py --width 1280 --height 720 --rate 1e8 --steps 12 --warmup 3 --repeats 3 --cpu-frames 0 --no-host-pass > gpurun_out/r02ad/bench_c5shape.json 2> gpurun_out/r02ad/bench_c5shape.err
python - <<'P'
import json
for f in ("bench_default","bench_c5shape"):
    d=json.load(open("gpurun_out/r02ad/%s.json"%f))
    print(f, d["value"], d["ms_per_step"], d.get("repeats"))
P
TAG=r02d bash tools/profile_bench.sh 2>&1 | tail -12
