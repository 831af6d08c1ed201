# round 2, GPU call: bench after the selection kernel became a template (LDS / device-memory bitmap)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ap
timeout 600 python bench.py --cpu-frames 0 --no-host-pass --repeats 3 > gpurun_out/r02ap/c3.json 2> gpurun_out/r02ap/c3.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r02ap/c3.json"))
print(d["value"], d["repeats"]["ms_per_step"])
for k,v in d["kernels"].items(): print("  ",k, v["avg_us"], v["launches"])
P
