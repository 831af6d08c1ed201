# round 2, GPU call 16: A/B of the prefetch sequence as a HIP graph with the new kernel sequence; trace
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02p
timeout 600 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "sae or replay" 2>&1 | tail -2
for g in 0 1 0 1; do
  if [ $g = 1 ]; then export ESVIO_FE_GRAPH=1; else unset ESVIO_FE_GRAPH; fi
  timeout 600 python bench.py --cpu-frames 0 --repeats 3 --no-host-pass --no-profile-pass > gpurun_out/r02p/b_graph$g.json 2> gpurun_out/r02p/b_graph$g.err
  python -c "
import json
d=json.load(open('gpurun_out/r02p/b_graph$g.json')); print('graph=$g', d['value'], d['repeats']['ms_per_step'])"
done
unset ESVIO_FE_GRAPH
ESVIO_FE_TRACE=1 timeout 600 python bench.py --steps 60 --warmup 6 --repeats 1 --cpu-frames 0 --no-profile-pass --no-host-pass > gpurun_out/r02p/trace.json 2> gpurun_out/r02p/trace.err
grep "esvio_fe trace" gpurun_out/r02p/trace.err | head -12
