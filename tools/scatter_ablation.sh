#!/bin/bash
# What k_tile_scatter costs without its parts (measurement builds; their results are WRONG by construction):
#   full       the kernel as shipped
#   nowrite    ranking and addresses computed, no record stored
#   coalesced  ranking done, every record stored at its stream position (the ideal 8 B/lane write)
#   norank     no ranking, scattered stores (ranks 0: fewer distinct lines than the real thing)
#   stream     no ranking, stores at the stream position: the kernel as a 16 B in / 8 B out copy
# Build here (CPU container):  bash tools/scatter_ablation.sh build
# Run on the GPU box:          gpurun -- 'bash tools/scatter_ablation.sh > gpurun_out/scatter_ablation.txt 2>&1'
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  python -m esvio_amd.build > /dev/null
  bash tools/build_variant.sh abl_nowrite -DESVIO_ABL_NOWRITE
  bash tools/build_variant.sh abl_coalesced -DESVIO_ABL_COALESCED
  bash tools/build_variant.sh abl_norank -DESVIO_ABL_NORANK
  bash tools/build_variant.sh abl_stream -DESVIO_ABL_NORANK -DESVIO_ABL_COALESCED
  exit 0
fi
for st in ${STREAMS:-scene poisson}; do
  echo "== full $st"; python tools/sae_microbench.py --stream $st --iters 24 2>&1 | grep -v "^$"
  for v in nowrite coalesced norank stream; do
    echo "== $v $st"; ESVIO_FE_LIB=tools/_bin/libesvio_fe_abl_$v.so python tools/sae_microbench.py --stream $st --iters 24 2>&1 | grep "scatter\|apply"
  done
done
