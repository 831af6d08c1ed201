# round 2, GPU call 27: the handle's own communicator (async exchange) against the torch.distributed mirror, one rank
cd $GRAFT_REPO_ROOT; timeout 900 python -m pytest tests/test_time_slice_gpu.py -m gpu -x -q 2>&1 | tail -12
mkdir -p gpurun_out/r02ac

for mode in "" "--force-dist" "--force-dist --torch-exchange"; do
  timeout 600 python bench.py --cpu-frames 0 --no-host-pass --no-profile-pass $mode > gpurun_out/r02ac/b.json 2> gpurun_out/r02ac/b.err
  python -c "
import json
d=json.load(open('gpurun_out/r02ac/b.json')); print('mode [$mode]', d['value'], d['repeats']['ms_per_step'], d['config'].get('track_exchange'))" || tail -5 gpurun_out/r02ac/b.err
done
