# round 2, GPU call 14: after the fe_api.cpp split
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02n
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02n/pytest.log
tail -3 gpurun_out/r02n/pytest.log
python -c "import __graft_entry__ as g; g.smoke()"
timeout 900 python bench.py --cpu-frames 10 --cpu-procs 0 > gpurun_out/r02n/bench_default.json 2> gpurun_out/r02n/bench_default.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r02n/bench_default.json"))
print(d["value"], d["ms_per_step"], d.get("repeats"))
P
