cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/prof_r01k_c5
rm -rf $O; mkdir -p $O
cd $R
CMD="python bench.py --width 1280 --height 720 --rate 1e8 --steps 12 --warmup 3 --cpu-frames 0 --no-profile-pass"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- $CMD > $O/trace.log 2>&1
T=$(find $O/trace -name "*.db" | head -1)
python tools/rocprof_summary.py --trace $T --out $R/gpurun_out/r01k_c5shape --note "round 1k, C5's sensor shape on one GPU: python bench.py --width 1280 --height 720 --rate 1e8 --steps 12 --warmup 3 --cpu-frames 0 --no-profile-pass (1280x720 stereo, 100 Mev/s per camera, 6.7 M events per step), MI355X"
grep value $O/trace.log | cut -c1-200
rm -rf $O/trace
