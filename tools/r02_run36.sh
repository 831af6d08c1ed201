# round 2, GPU call 22: benches + rocprofv3 passes (profiles r02e), collective path with one rank
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02am
timeout 900 python bench.py > gpurun_out/r02am/bench_default.json 2> gpurun_out/r02am/bench_default.err
timeout 600 python bench.py --force-dist --cpu-frames 0 --no-host-pass --no-profile-pass > gpurun_out/r02am/bench_forcedist.json 2> gpurun_out/r02am/bench_forcedist.err
timeout 600 python bench.py --width 1280 --height 720 --rate 1e8 --steps 12 --warmup 3 --repeats 3 --cpu-frames 0 --no-host-pass > gpurun_out/r02am/bench_c5shape.json 2> gpurun_out/r02am/bench_c5shape.err
python - <<'P'
import json
for f in ("bench_default","bench_forcedist","bench_c5shape"):
    try:
        d=json.load(open("gpurun_out/r02am/%s.json"%f))
        print(f, d["value"], d["ms_per_step"], d.get("repeats"), d["config"].get("track_exchange"))
    except Exception as e:
        print(f, "FAILED", e)
P
tail -3 gpurun_out/r02am/bench_forcedist.err
TAG=r02e bash tools/profile_bench.sh 2>&1 | tail -12
