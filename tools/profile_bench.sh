# rocprofv3 passes behind profiles/<tag>_*: the default bench (C3) in the replay schedule (with the
# per-cycle device timeline) and with one batch in flight (clean per-kernel durations), the two PMC
# passes (separate runs, never combined with tracing), and the C5 sensor shape.   usage: TAG=r02a bash tools/profile_bench.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r02}
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd $R
CMD="python bench.py --steps 40 --warmup 5 --repeats 1 --cpu-frames 0 --no-profile-pass --no-host-pass --no-sae-pass"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace -o trace -- $CMD > $O/trace.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/plain -o plain -- $CMD --no-pipeline --no-chain > $O/plain.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o fetch -- $CMD --no-pipeline --no-chain > $O/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/write -o write -- $CMD --no-pipeline --no-chain > $O/write.log 2>&1
T=$(find $O/trace -name "*.db" | head -1); P=$(find $O/plain -name "*.db" | head -1)
F=$(find $O/fetch -name "*.db" | head -1); W=$(find $O/write -name "*.db" | head -1)
python tools/rocprof_summary.py --trace $P --pmc FETCH_SIZE=$F --pmc WRITE_SIZE=$W --out $R/gpurun_out/${TAG}_bench_c3 --note "$TAG, one batch in flight (--no-pipeline --no-chain: no speculative / chained / lazy launches, clean kernel durations): $CMD --no-pipeline --no-chain (C3 640x480 stereo), MI355X; FETCH_SIZE/WRITE_SIZE in KB per dispatch, separate --pmc passes of the same command" > /dev/null
python tools/rocprof_summary.py --trace $T --out $R/gpurun_out/${TAG}_bench_c3_replay --note "$TAG, replay schedule (3 batches announced ahead, lazy): $CMD; the k_lk average contains the time speculative / chained launches wait for their inputs" > /dev/null
python tools/timeline.py $T > $R/gpurun_out/${TAG}_bench_c3_replay_timeline.txt 2>&1
grep -h '"value"' $O/trace.log $O/plain.log | cut -c1-160
# the DEFAULT schedule at the headline rate over a long timed region (200 steps, one pass): union-busy per
# step from the trace against the ms_per_step the same process prints
CMDH="python bench.py --steps 200 --warmup 20 --repeats 1 --cpu-frames 0 --no-profile-pass --no-host-pass --no-sae-pass"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/head -o head -- $CMDH > $O/head.log 2>&1
TH=$(find $O/head -name "*.db" | head -1)
python tools/rocprof_summary.py --trace $TH --out $R/gpurun_out/${TAG}_bench_headline --note "$TAG, the default bench schedule over 200 timed steps: $CMDH; ms_per_step printed by the same process: $(grep -h -o '"ms_per_step": [0-9.]*' $O/head.log | head -1)" > /dev/null
python tools/timeline.py $TH > $R/gpurun_out/${TAG}_bench_headline_timeline.txt 2>&1
grep -h '"value"' $O/head.log | cut -c1-200; head -8 $R/gpurun_out/${TAG}_bench_headline_timeline.txt
rm -rf $O/head
CMD5="python bench.py --width 1280 --height 720 --rate 1e8 --steps 12 --warmup 3 --repeats 1 --cpu-frames 0 --no-profile-pass --no-host-pass --no-sae-pass"
timeout 900 rocprofv3 --kernel-trace --stats -d $O/c5 -o c5 -- $CMD5 > $O/c5.log 2>&1
timeout 900 rocprofv3 --pmc FETCH_SIZE -d $O/c5f -o c5f -- $CMD5 > $O/c5f.log 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE -d $O/c5w -o c5w -- $CMD5 > $O/c5w.log 2>&1
T5=$(find $O/c5 -name "*.db" | head -1); F5=$(find $O/c5f -name "*.db" | head -1); W5=$(find $O/c5w -name "*.db" | head -1)
python tools/rocprof_summary.py --trace $T5 --pmc FETCH_SIZE=$F5 --pmc WRITE_SIZE=$W5 --out $R/gpurun_out/${TAG}_c5shape --note "$TAG, C5's sensor shape on one GPU: $CMD5 (1280x720 stereo, 100 Mev/s per camera, 6.7 M events per step), MI355X" > /dev/null
grep -h '"value"' $O/c5.log | cut -c1-160
# the event-proportional chain alone at C5's batch size (tools/sae_microbench.py), scene and uniform stream
for st in scene poisson; do
  CMDS="python tools/sae_microbench.py --stream $st --iters 24"
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/sae_$st -o sae -- $CMDS > $O/sae_$st.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE -d $O/saef_$st -o saef -- $CMDS > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE -d $O/saew_$st -o saew -- $CMDS > /dev/null 2>&1
  TS=$(find $O/sae_$st -name "*.db" | head -1); FS=$(find $O/saef_$st -name "*.db" | head -1); WS=$(find $O/saew_$st -name "*.db" | head -1)
  python tools/rocprof_summary.py --trace $TS --pmc FETCH_SIZE=$FS --pmc WRITE_SIZE=$WS --out $R/gpurun_out/${TAG}_sae_microbench_$st --note "$TAG: $CMDS (1280x720 stereo, 6.7 M events per batch, device-resident; createSAE_left/right only), MI355X" > /dev/null
  grep -h "events per batch" $O/sae_$st.log; grep "k_tile" $R/gpurun_out/${TAG}_sae_microbench_$st.md | cut -c1-120
  rm -rf $O/sae_$st $O/saef_$st $O/saew_$st
done
head -30 $R/gpurun_out/${TAG}_bench_c3_replay_timeline.txt
rm -rf $O/trace $O/plain $O/fetch $O/write $O/c5 $O/c5f $O/c5w
