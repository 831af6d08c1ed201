cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_c5_full_rate_gpu.py -x -q -m gpu -k "sae or c5 or record or motion" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for st in scene poisson; do
  rm -rf /tmp/p_$st
  rocprofv3 --kernel-trace --stats -d /tmp/p_$st -o sae -- python $R/tools/sae_microbench.py --stream $st --iters 24 > /tmp/log_$st.txt 2>&1
  db=$(find /tmp/p_$st -name '*.db' | head -1)
  python $R/tools/rocprof_summary.py --trace $db --out $R/gpurun_out/sae_$st > /dev/null 2>&1
  echo "== $st"; grep "k_tile" $R/gpurun_out/sae_$st.md | awk -F'|' '{printf "  %-34s avg %s min %s\n", $2, $5, $6}'
done
rm -rf /tmp/p_c3
rocprofv3 --kernel-trace --stats -d /tmp/p_c3 -o sae -- python $R/tools/sae_microbench.py --width 640 --height 480 --rate 5e6 --stream scene --iters 40 > /tmp/log_c3.txt 2>&1
db=$(find /tmp/p_c3 -name '*.db' | head -1)
python $R/tools/rocprof_summary.py --trace $db --out $R/gpurun_out/sae_c3 > /dev/null 2>&1
echo "== c3"; grep "k_tile" $R/gpurun_out/sae_c3.md | awk -F'|' '{printf "  %-34s avg %s min %s\n", $2, $5, $6}'
