# round 2, first GPU call: full -m gpu suite (with the new C4/C5/LK-order tests), default bench, C5-shape bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02a
timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | tail -60 > gpurun_out/r02a/pytest.log
tail -5 gpurun_out/r02a/pytest.log
timeout 600 python bench.py > gpurun_out/r02a/bench_default.json 2> gpurun_out/r02a/bench_default.err
cut -c1-900 gpurun_out/r02a/bench_default.json
timeout 600 python bench.py --width 1280 --height 720 --rate 1e8 --steps 12 --warmup 3 --cpu-frames 0 > gpurun_out/r02a/bench_c5shape.json 2> gpurun_out/r02a/bench_c5shape.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r02a/bench_c5shape.json"))
print(d["value"], d["ms_per_step"])
for k,v in d["kernels"].items(): print(k, v["avg_us"], v["launches"], v["achieved_GBs"])
P
