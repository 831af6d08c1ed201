# round 2, GPU call 25: the GPU suite five times over (flakiness check)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02y
for i in 1 2 3 4 5; do
  timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider 2>&1 | tail -2
done
