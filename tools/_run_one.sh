cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_c5_full_rate_gpu.py -x -q -m gpu -k "sae or c5 or record or motion" 2>&1 | tail -4
STREAMS="scene poisson" bash tools/_run_ab.sh
