# round 2: does binding the calling thread to its L3 domain steady the step time? A/B, 3 runs each
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02at
for rep in 1 2 3; do
for mode in pin nopin; do
  if [ $mode = nopin ]; then F="--no-pin-caller"; else F=""; fi
  timeout 600 python bench.py $F --cpu-frames 0 --no-host-pass --no-profile-pass --repeats 3 > gpurun_out/r02at/${mode}_$rep.json 2> gpurun_out/r02at/${mode}_$rep.err
  python - <<P
import json
d=json.load(open("gpurun_out/r02at/${mode}_$rep.json")); print("$mode", $rep, d["repeats"]["ms_per_step"], [h["us_per_step"] for h in d["host_ransac"]])
P
done
done
