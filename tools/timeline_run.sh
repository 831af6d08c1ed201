cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tl
rm -rf $O; mkdir -p $O
cd $R
env ${GRAPH:+ESVIO_FE_GRAPH=1} timeout 600 rocprofv3 --kernel-trace -d $O/trace -o trace -- python bench.py --steps 120 --warmup 10 --cpu-frames 0 --no-profile-pass --host-threads 6 > $O/trace.log 2>&1
T=$(find $O/trace -name "*.db" | head -1)
python tools/timeline.py $T > $O/timeline.txt 2>&1
tail -c 300 $O/trace.log | cut -c1-200
cat $O/timeline.txt
rm -rf $O/trace
