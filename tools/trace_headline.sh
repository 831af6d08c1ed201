# rocprofv3 kernel trace of the default bench schedule + the per-cycle timeline (tools/timeline.py): usage TAG=x bash tools/trace_headline.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; TAG=${TAG:-t}; O=$R/gpurun_out/prof_$TAG; rm -rf $O; mkdir -p $O; cd $R
CMDH="python bench.py --steps 200 --warmup 20 --repeats 1 --cpu-frames 0 --no-profile-pass --no-host-pass --no-sae-pass"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/head -o head -- $CMDH > $O/head.log 2>&1
TH=$(find $O/head -name "*.db" | head -1)
python tools/timeline.py $TH > $R/gpurun_out/${TAG}_timeline.txt 2>&1
grep -h -o '"ms_per_step": [0-9.]*' $O/head.log | head -1; head -60 $R/gpurun_out/${TAG}_timeline.txt
rm -rf $O
