# round 2, GPU call 23: fused k_ts_pyr against the unfused kernels at C5's size (A/B)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02w
for nf in 0 1; do
  if [ $nf = 1 ]; then export ESVIO_FE_NO_FUSE=1; else unset ESVIO_FE_NO_FUSE; fi
  timeout 600 python bench.py --width 1280 --height 720 --rate 1e8 --steps 12 --warmup 3 --repeats 3 --cpu-frames 0 --no-host-pass > gpurun_out/r02w/b_nofuse$nf.json 2> gpurun_out/r02w/b_nofuse$nf.err
  python -c "
import json
d=json.load(open('gpurun_out/r02w/b_nofuse$nf.json')); k=d['kernels']; print('nofuse=$nf', d['value'], d['repeats']['ms_per_step'], {x:(k[x]['avg_us'],k[x]['launches']) for x in ('k_time_surface','k_pyr_down','k_pyr_pad','k_scharr') if x in k})"
done
