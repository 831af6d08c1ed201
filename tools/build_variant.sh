#!/bin/bash
# build an experimental variant of the library: tools/build_variant.sh NAME -DFLAG...  ->  tools/_bin/libesvio_fe_NAME.so
# (run with ESVIO_FE_LIB=tools/_bin/libesvio_fe_NAME.so)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/_bin
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fno-math-errno "$@" \
  -x hip -c esvio_amd/csrc/fe_kernels.hip -o tools/_bin/fe_kernels_$name.o
objs=$(ls esvio_amd/build/*.o | grep -v fe_kernels)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC tools/_bin/fe_kernels_$name.o $objs -Wl,-rpath,/opt/rocm/lib -ldl -o tools/_bin/libesvio_fe_$name.so
rm tools/_bin/fe_kernels_$name.o
echo tools/_bin/libesvio_fe_$name.so
