# round 2, GPU call 24: k_select with one bitmap look per step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02x
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02x/pytest.log
tail -3 gpurun_out/r02x/pytest.log
timeout 300 python tools/select_microbench.py 2>&1 | tail -8
timeout 900 python bench.py --cpu-frames 10 --cpu-procs 0 > gpurun_out/r02x/bench_default.json 2> gpurun_out/r02x/bench_default.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r02x/bench_default.json"))
print(d["value"], d["ms_per_step"], d["repeats"]["ms_per_step"])
for k,v in d["kernels"].items(): print("  ",k, v["avg_us"], v["launches"], (d.get("kernels_replay_schedule") or {}).get(k,{}).get("avg_us"))
P
