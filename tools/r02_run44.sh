# round 2, GPU call: final verification of the build — GPU suite, smoke, benches, rocprofv3 passes (profiles r02g)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02au
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 > gpurun_out/r02au/pytest.log
tail -2 gpurun_out/r02au/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 900 python bench.py > gpurun_out/r02au/bench_default.json 2> gpurun_out/r02au/bench_default.err
timeout 600 python bench.py --equalize 1 --cpu-frames 0 --no-host-pass > gpurun_out/r02au/bench_equalize.json 2> gpurun_out/r02au/bench_equalize.err
timeout 600 python bench.py --width 1280 --height 720 --rate 1e8 --steps 12 --warmup 3 --repeats 3 --cpu-frames 0 --no-host-pass > gpurun_out/r02au/bench_c5shape.json 2> gpurun_out/r02au/bench_c5shape.err
python - <<'P'
import json
for f in ("bench_default","bench_equalize","bench_c5shape"):
    try:
        d=json.load(open("gpurun_out/r02au/%s.json"%f))
        print(f, d["value"], d["ms_per_step"], d["repeats"]["ms_per_step"], (d.get("float_order_lk") or {}).get("ms_per_step"))
    except Exception as e:
        print(f, "FAILED", e)
P
TAG=r02g bash tools/profile_bench.sh 2>&1 | tail -3
