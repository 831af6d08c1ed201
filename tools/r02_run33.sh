# round 2, GPU call 18: kernel traces of the old (_ab) and the current tree, same command, to see where the step time goes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02ai
rm -rf $O; mkdir -p $O
CMD="python bench.py --steps 60 --warmup 6 --repeats 3 --cpu-frames 0 --no-profile-pass --no-host-pass"
for side in old new; do
  if [ $side = old ]; then dir=$R/_ab; else dir=$R; fi
  cd $dir
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/$side -o t -- $CMD > $O/$side.log 2>&1
  T=$(find $O/$side -name "*.db" | head -1)
  python $R/tools/rocprof_summary.py --trace $T --out $O/${side}_summary > /dev/null
  python $R/tools/timeline.py $T > $O/${side}_timeline.txt 2>&1
  grep -h '"value"' $O/$side.log | cut -c1-300
  head -8 $O/${side}_timeline.txt
  rm -rf $O/$side
done
