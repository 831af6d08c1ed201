# round 2, GPU call 26: device plane sets + RCCL with one rank (test), bench.py through the collective path with one rank
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02z
timeout 900 python -m pytest tests/test_time_slice_gpu.py -m gpu -x -q 2>&1 | tail -15
for split in rigs time; do
  timeout 600 python bench.py --steps 20 --warmup 5 --repeats 2 --cpu-frames 0 --no-host-pass --no-profile-pass --force-dist --split $split > gpurun_out/r02z/b_force_$split.json 2> gpurun_out/r02z/b_force_$split.err
  echo "force-dist $split rc=$?"; cut -c1-330 gpurun_out/r02z/b_force_$split.json; tail -3 gpurun_out/r02z/b_force_$split.err
done
