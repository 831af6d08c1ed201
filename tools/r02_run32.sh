# round 2, GPU call 17: A/B on one box — tree before the Jacobi null space (_ab) against the current one
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ah
for rep in 1 2; do
for side in old new; do
  if [ $side = old ]; then dir=_ab; else dir=.; fi
  (cd $dir && timeout 600 python bench.py --cpu-frames 0 --no-host-pass --repeats 3 > $GRAFT_REPO_ROOT/gpurun_out/r02ah/bench_${side}_$rep.json 2> $GRAFT_REPO_ROOT/gpurun_out/r02ah/bench_${side}_$rep.err)
  python - <<P
import json
d=json.load(open("gpurun_out/r02ah/bench_${side}_$rep.json"))
print("$side", $rep, d["value"], d["repeats"]["ms_per_step"], d.get("host_ransac"))
P
done
done
