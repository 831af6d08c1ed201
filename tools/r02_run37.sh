# round 2, GPU call 24: host thread sweep with the lane RANSAC / pooled LMedS
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02an
for th in 8 12 6 10 14 4 8; do
timeout 600 python bench.py --host-threads $th --cpu-frames 0 --no-host-pass --no-profile-pass --repeats 3 > gpurun_out/r02an/bench_th$th.json 2> gpurun_out/r02an/bench_th$th.err
python - <<P
import json
d=json.load(open("gpurun_out/r02an/bench_th$th.json"))
print($th, d["repeats"]["ms_per_step"], [ (h["mean_us"], h["lmeds_mean_us"], h["us_per_step"]) for h in d["host_ransac"]])
P
done
nproc; cat /sys/fs/cgroup/cpu.max
