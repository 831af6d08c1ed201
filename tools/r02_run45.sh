# round 2, GPU call: inlined hypot in the lane solver — RANSAC cost, thread sweep
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02av
g++ -O3 -std=c++17 -ffp-contract=off -fno-math-errno -pthread tools/ransac_bench.cpp esvio_amd/csrc/fe_host.cpp -Iinclude -o /tmp/ransac_bench && /tmp/ransac_bench 160 0.37
python -m pytest tests -m gpu -x -q -k "ransac or track_event_end_to_end or replay or shipped or abi" 2>&1 | tail -2
for th in 8 4 2 1; do
python bench.py --host-threads $th --cpu-frames 0 --no-host-pass --no-profile-pass --repeats 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print($th, d['repeats']['ms_per_step'], [(h['mean_us'], h['lmeds_mean_us'], h['us_per_step']) for h in d['host_ransac']])"
done
