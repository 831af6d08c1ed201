# round 2, GPU call 8: N>1 dry runs of bench.py on one GPU (gloo), host trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02h
for split in rigs camera time; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 --repeats 1 --cpu-frames 0 --split $split --dist-backend gloo > gpurun_out/r02h/bench_2_$split.json 2> gpurun_out/r02h/bench_2_$split.err
  echo "split $split rc=$?"; tail -c 600 gpurun_out/r02h/bench_2_$split.json | cut -c1-600; tail -3 gpurun_out/r02h/bench_2_$split.err
done
ESVIO_FE_TRACE=1 timeout 600 python bench.py --steps 60 --warmup 6 --repeats 1 --cpu-frames 0 --no-profile-pass --no-host-pass > gpurun_out/r02h/trace.json 2> gpurun_out/r02h/trace.err
tail -25 gpurun_out/r02h/trace.err
