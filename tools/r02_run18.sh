# round 2, GPU call 18: k_tile_apply block size A/B at C5's batch size
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02r
for t in 1024 512 256; do
  export ESVIO_FE_APPLY_THREADS=$t
  timeout 600 python bench.py --width 1280 --height 720 --rate 1e8 --steps 12 --warmup 3 --repeats 3 --cpu-frames 0 --no-host-pass > gpurun_out/r02r/b_$t.json 2> gpurun_out/r02r/b_$t.err
  python -c "
import json
d=json.load(open('gpurun_out/r02r/b_$t.json')); k=d['kernels']; print('threads=$t', d['value'], d['repeats']['ms_per_step'], 'apply', k['k_sae_apply']['avg_us'])"
done
