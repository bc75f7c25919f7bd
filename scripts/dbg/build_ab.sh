#!/bin/bash
# A/B builds of be_solve.hip with different -D switches -> abbuild/libdvins_<name>.so (git-ignored, travels with gpurun); use with DVINS_HIP_LIB=$PWD/abbuild/libdvins_<name>.so
#   usage: scripts/dbg/build_ab.sh name "-DMF_WAVE_T=0 -DMF_BS_MM=0" [name2 "flags2" ...]
set -e
cd "$(dirname "$0")/../../dynamic_vins_amd/csrc"
make -s -j8
mkdir -p ../../abbuild/obj
for f in *.hip; do cp build/${f%.hip}.o ../../abbuild/obj/${f%.hip}.o; done
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  src=be_solve; case $name in marg*) src=be_marg;; eval*) src=be_eval;; esac          # a name that starts with "marg" rebuilds be_marg.hip instead of be_solve.hip
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -w $flags -c -o ../../abbuild/obj/ab_${src}_$name.o $src.hip
  objs=$(ls ../../abbuild/obj/*.o | grep -v "/ab_\|/be_solve_\|/$src.o")
  /opt/rocm/bin/hipcc -O3 -fPIC --offload-arch=gfx950 -shared -o ../../abbuild/libdvins_$name.so $objs ../../abbuild/obj/ab_${src}_$name.o -ldl
done
