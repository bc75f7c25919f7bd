#!/bin/bash
# stamped build of the library (be_solve phase stamps: -DBE_SOLVE_TS; `build_ts.sh marg`: the marginalization's, -DBE_MARG_TS) -> dbgbuild/libdvins_hip_ts.so; use with DVINS_HIP_LIB=$PWD/dbgbuild/libdvins_hip_ts.so
set -e
cd "$(dirname "$0")/../../dynamic_vins_amd/csrc"
make -s -j8
mkdir -p ../../dbgbuild
for f in *.hip; do cp build/${f%.hip}.o ../../dbgbuild/${f%.hip}.o; done
if [ "$1" = marg ]; then /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DBE_MARG_TS -c -o ../../dbgbuild/be_marg.o be_marg.hip
else /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DBE_SOLVE_TS -c -o ../../dbgbuild/be_solve.o be_solve.hip; fi
/opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o ../../dbgbuild/libdvins_hip_ts.so ../../dbgbuild/*.o -ldl
