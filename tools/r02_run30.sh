# round 2, GPU call 15: 7-point systems solved kLanes at a time; host thread sweep
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02af
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02af/pytest.log
tail -3 gpurun_out/r02af/pytest.log
g++ -O3 -std=c++17 -ffp-contract=off -fno-math-errno -pthread tools/ransac_bench.cpp esvio_amd/csrc/fe_host.cpp -Iinclude -o /tmp/ransac_bench && /tmp/ransac_bench 160 0.37 > gpurun_out/r02af/ransac_bench.txt 2>&1
cat gpurun_out/r02af/ransac_bench.txt
for th in 8 4 12 16; do
timeout 600 python bench.py --host-threads $th --cpu-frames 0 --no-host-pass --repeats 3 > gpurun_out/r02af/bench_th$th.json 2> gpurun_out/r02af/bench_th$th.err
python - <<P
import json
d=json.load(open("gpurun_out/r02af/bench_th$th.json"))
print($th, d["value"], d["ms_per_step"], d.get("repeats"))
P
done
