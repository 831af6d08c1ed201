# round 2, GPU call 21: GPU suite on the build with cv::SVD's null space + lane/pool RANSAC and LMedS
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02al
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r02al/pytest.log
tail -3 gpurun_out/r02al/pytest.log
g++ -O3 -std=c++17 -ffp-contract=off -fno-math-errno -pthread tools/ransac_bench.cpp esvio_amd/csrc/fe_host.cpp -Iinclude -o /tmp/ransac_bench && /tmp/ransac_bench 160 0.37 > gpurun_out/r02al/ransac_bench.txt 2>&1
cat gpurun_out/r02al/ransac_bench.txt
