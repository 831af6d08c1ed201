# round 2, GPU call: smoke + benches + rocprofv3 passes (profiles r02f)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02as
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/r02as/bench_default.json 2> gpurun_out/r02as/bench_default.err
timeout 600 python bench.py --equalize 1 --cpu-frames 0 --no-host-pass > gpurun_out/r02as/bench_equalize.json 2> gpurun_out/r02as/bench_equalize.err
timeout 600 python bench.py --width 1280 --height 720 --rate 1e8 --steps 12 --warmup 3 --repeats 3 --cpu-frames 0 --no-host-pass > gpurun_out/r02as/bench_c5shape.json 2> gpurun_out/r02as/bench_c5shape.err
python - <<'P'
import json
for f in ("bench_default","bench_equalize","bench_c5shape"):
    try:
        d=json.load(open("gpurun_out/r02as/%s.json"%f))
        print(f, d["value"], d["ms_per_step"], d["repeats"]["ms_per_step"], d["roofline"]["traffic_source"] if d.get("roofline") else None)
    except Exception as e:
        print(f, "FAILED", e)
P
TAG=r02f bash tools/profile_bench.sh 2>&1 | tail -6
