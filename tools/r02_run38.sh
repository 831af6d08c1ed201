# round 2, GPU call 25: partition half of a prefetched batch's SAE update on its own stream — A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ao
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/r02ao/pytest.log
tail -3 gpurun_out/r02ao/pytest.log
C5="--width 1280 --height 720 --rate 1e8 --steps 12 --warmup 3 --repeats 3 --cpu-frames 0 --no-host-pass --no-profile-pass"
C3="--cpu-frames 0 --no-host-pass --no-profile-pass --repeats 3"
for rep in 1 2; do
for mode in on off; do
  if [ $mode = off ]; then export ESVIO_FE_NO_PART_OVERLAP=1; else unset ESVIO_FE_NO_PART_OVERLAP; fi
  timeout 600 python bench.py $C5 > gpurun_out/r02ao/c5_${mode}_$rep.json 2> gpurun_out/r02ao/c5_${mode}_$rep.err
  timeout 600 python bench.py $C3 > gpurun_out/r02ao/c3_${mode}_$rep.json 2> gpurun_out/r02ao/c3_${mode}_$rep.err
  python - <<P
import json
for w in ("c5","c3"):
    try:
        d=json.load(open("gpurun_out/r02ao/%s_${mode}_$rep.json"%w)); print("$mode", $rep, w, d["value"], d["repeats"]["ms_per_step"])
    except Exception as e: print("$mode", w, "FAILED", e)
P
done
done
