# round 2, GPU call 17: does the largest bucket want a CU of its own?  (LDS request A/B for k_tile_apply<1024>)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02q
for kib in 0 90 60; do
  export ESVIO_FE_APPLY_LDS_KIB=$kib
  timeout 600 python bench.py --width 1280 --height 720 --rate 1e8 --steps 12 --warmup 3 --repeats 3 --cpu-frames 0 --no-host-pass > gpurun_out/r02q/b_$kib.json 2> gpurun_out/r02q/b_$kib.err
  python -c "
import json
d=json.load(open('gpurun_out/r02q/b_$kib.json')); k=d['kernels']; print('lds_kib=$kib', d['value'], d['repeats']['ms_per_step'], 'apply', k['k_sae_apply']['avg_us'])"
done
