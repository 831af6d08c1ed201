cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r01k
rm -rf $O; mkdir -p $O
cd $R
CMD="python bench.py --steps 40 --warmup 5 --cpu-frames 0 --no-profile-pass"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- $CMD > $O/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o fetch -- $CMD > $O/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/write -o write -- $CMD > $O/write.log 2>&1
find $O -name "*.db" | head
T=$(find $O/trace -name "*.db" | head -1); F=$(find $O/fetch -name "*.db" | head -1); W=$(find $O/write -name "*.db" | head -1)
python tools/rocprof_summary.py --trace $T --pmc FETCH_SIZE=$F --pmc WRITE_SIZE=$W --out $R/gpurun_out/r01k_bench_c3 --note "round 1k (as 1h + fused k_ts_pyr and k_pad_scharr, k_select disc threshold; 1h = 1g + lazy right tail, RANSAC helper threads, stereo LK stream, chained temporal LK at prefetch depth 3, candidate dedup, kept-point discs stamped by k_select), python bench.py --steps 40 --warmup 5 --cpu-frames 0 --no-profile-pass (C3 640x480 stereo, ego scene), MI355X; FETCH_SIZE/WRITE_SIZE in KB per dispatch, separate --pmc passes"
grep value $O/trace.log | cut -c1-200
python bench.py > $R/gpurun_out/r01k_bench_default.json 2> $R/gpurun_out/r01k_bench_default.err; tail -c 1500 $R/gpurun_out/r01k_bench_default.json
