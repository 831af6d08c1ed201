# kernel trace of trackImage at 346x260
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02ar
rm -rf $O; mkdir -p $O
cd $R
cat > /tmp/img346.py <<'P'
import sys, time
sys.path.insert(0, ".")
from esvio_amd import frontend as FE
from esvio_amd.synth import ImageStream
W,H=346,260
s=ImageStream(W,H,velocity=(4,-2),disparity=12,seed=3)
frames=[s.next_frame() for _ in range(24)]
ft=FE.FeatureTracker(FE.make_config(W,H,max_cnt=150,min_dist=10,flow_back=1))
for k,(L,R,t) in enumerate(frames):
    t0=time.perf_counter(); ft.trackImage(t,L,R,k%2==0); print(k, round((time.perf_counter()-t0)*1e3,3), len(ft.ids))
P
timeout 300 rocprofv3 --kernel-trace --stats -d $O/t -o t -- python /tmp/img346.py > $O/t.log 2>&1
T=$(find $O/t -name "*.db" | head -1)
python tools/rocprof_summary.py --trace $T --out $O/img346 > /dev/null
tail -26 $O/t.log | head -24
head -24 $O/img346.md
rm -rf $O/t
