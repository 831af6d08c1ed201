# A/B of library variants under gpurun_ab/ through the SAE microbench (rocprofv3 kernel trace).  Build the
# variants here (hipcc with the flags of esvio_amd/build.py + -DSOME_MACRO=n -o gpurun_ab/lib_<name>.so), then
#   gpurun -- 'bash tools/ab_sae.sh'            (STREAMS="scene poisson" by default)
# The library in place is restored at the end; gpurun_ab/ is scratch (git-ignored).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/esvio_amd/libesvio_fe.so /tmp/lib_orig.so
for lib in $R/gpurun_ab/lib_*.so; do
  v=$(basename $lib .so)
  cp $lib $R/esvio_amd/libesvio_fe.so
  for st in ${STREAMS:-scene poisson}; do
    rm -rf /tmp/p_$v$st
    rocprofv3 --kernel-trace --stats -d /tmp/p_$v$st -o sae -- python $R/tools/sae_microbench.py --stream $st --iters 24 > /tmp/log_$v$st.txt 2>&1
    db=$(find /tmp/p_$v$st -name '*.db' | head -1)
    python $R/tools/rocprof_summary.py --trace $db --out /tmp/sum_$v$st > /dev/null 2>&1
    echo "== $v $st"; grep "k_tile" /tmp/sum_$v$st.md | awk -F'|' '{printf "  %-34s avg %s min %s\n", $2, $5, $6}'
  done
done
cp /tmp/lib_orig.so $R/esvio_amd/libesvio_fe.so
