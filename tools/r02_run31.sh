# round 2, GPU call 16: host RANSAC statistics of the bench stream
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ag
for th in 8 1; do
timeout 600 python bench.py --host-threads $th --cpu-frames 0 --no-host-pass --repeats 3 > gpurun_out/r02ag/bench_th$th.json 2> gpurun_out/r02ag/bench_th$th.err
python - <<P
import json
d=json.load(open("gpurun_out/r02ag/bench_th$th.json"))
print($th, d["value"], d["ms_per_step"], d.get("repeats")["ms_per_step"], d["host_ransac"])
P
done
