# round 2: kernel trace of the equalize: 1 path (esio_DSEC's shipped setting)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02aq
rm -rf $O; mkdir -p $O
cd $R
CMD="python bench.py --equalize 1 --steps 40 --warmup 5 --repeats 1 --cpu-frames 0 --no-profile-pass --no-host-pass"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/t -o t -- $CMD > $O/t.log 2>&1
T=$(find $O/t -name "*.db" | head -1)
python tools/rocprof_summary.py --trace $T --out $O/eq1 > /dev/null
head -30 $O/eq1.md
rm -rf $O/t
