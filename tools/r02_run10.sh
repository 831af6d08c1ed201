# round 2, GPU call 10: is the candidate dedup worth its atomics?  (A/B)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02j
for dd in 0 1; do
  for cfg in "c3:" "c5:--width 1280 --height 720 --rate 1e8 --steps 12 --warmup 3"; do
    name=${cfg%%:*}; args=${cfg#*:}
    if [ $dd = 1 ]; then export ESVIO_FE_NO_DEDUP=1; else unset ESVIO_FE_NO_DEDUP; fi
    timeout 600 python bench.py $args --repeats 1 --cpu-frames 0 --no-host-pass > gpurun_out/r02j/b_${name}_nodedup$dd.json 2> gpurun_out/r02j/b_${name}_nodedup$dd.err
    python - <<P
import json
d=json.load(open("gpurun_out/r02j/b_${name}_nodedup$dd.json"))
k=d["kernels"]
print("$name nodedup=$dd", d["value"], d["ms_per_step"], {x:(k[x]["avg_us"],k[x]["launches"]) for x in ("k_sae_keys","k_arc","k_arc_map","k_compact","k_select") if x in k})
P
  done
done
