# round 2, GPU call 20: host phase trace (ESVIO_FE_TRACE) of the current tree; LMedS through the lanes/pool
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02ak
rm -rf $O; mkdir -p $O
CMD="python bench.py --steps 60 --warmup 6 --repeats 3 --cpu-frames 0 --no-profile-pass --no-host-pass"
cd $R
ESVIO_FE_TRACE=1 timeout 600 $CMD > $O/trace.json 2> $O/trace.err
grep "esvio_fe trace" $O/trace.err | cut -c1-700
timeout 600 $CMD > $O/new.json 2> $O/new.err
python -c "
import json; d=json.load(open('$O/new.json')); print(d['repeats']['ms_per_step'], d['host_ransac'])"
cd $R/_ab; timeout 600 $CMD > $O/old.json 2> $O/old.err
python -c "
import json; d=json.load(open('$O/old.json')); print(d['repeats']['ms_per_step'])"
