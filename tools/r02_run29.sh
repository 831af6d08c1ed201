# round 2, GPU call 14: run7Point's null space by cv::SVD's Jacobi route (host RANSAC cost), tests + bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02ae
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02ae/pytest.log
tail -3 gpurun_out/r02ae/pytest.log
g++ -O3 -std=c++17 -ffp-contract=off -pthread tools/ransac_bench.cpp esvio_amd/csrc/fe_host.cpp -Iinclude -o /tmp/ransac_bench && /tmp/ransac_bench 160 0.37 > gpurun_out/r02ae/ransac_bench.txt 2>&1
cat gpurun_out/r02ae/ransac_bench.txt
timeout 900 python bench.py > gpurun_out/r02ae/bench_default.json 2> gpurun_out/r02ae/bench_default.err
python - <<'P'
import json
d=json.load(open("gpurun_out/r02ae/bench_default.json"))
print(d["value"], d["ms_per_step"], d.get("repeats"), d.get("host_resident_events"))
P
lscpu | grep -i "model name\|^CPU(s)"
