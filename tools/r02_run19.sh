# round 2, GPU call 19: k_tile_apply turn length A/B at C5's batch size (two library builds)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02s
export ESVIO_FE_APPLY_THREADS=512
for lib in turn2 turn8; do
  cp gpurun_out_libs/libesvio_fe_$lib.so esvio_amd/libesvio_fe.so
  timeout 600 python bench.py --width 1280 --height 720 --rate 1e8 --steps 12 --warmup 3 --repeats 3 --cpu-frames 0 --no-host-pass > gpurun_out/r02s/b_$lib.json 2> gpurun_out/r02s/b_$lib.err
  python -c "
import json
d=json.load(open('gpurun_out/r02s/b_$lib.json')); k=d['kernels']; print('$lib', d['value'], d['repeats']['ms_per_step'], 'apply', k['k_sae_apply']['avg_us'])"
done
