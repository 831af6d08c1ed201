# the whole -m gpu suite + the driver's bench line (what the round-end run does), into gpurun_out/
TAG=${TAG:-r03}
python -m pytest tests -q -m gpu -x 2>&1 | tail -8 > gpurun_out/${TAG}_gputests.txt
cat gpurun_out/${TAG}_gputests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_driver.json 2> gpurun_out/${TAG}_bench_driver.err
python - <<PY
import json
d=json.load(open("gpurun_out/${TAG}_bench_driver.json"))
print({k:d[k] for k in ("value","ms_per_step","host_resident_events","float_order_lk","repeats")})
print(d["roofline"]); print(d["cpu_baseline"]["value"], d["cpu_baseline"].get("all_cores"))
PY
