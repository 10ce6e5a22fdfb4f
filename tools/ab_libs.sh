#!/bin/bash
# Builds one library per switch of policy.hip (same-box A/B of a kernel change): imitation_amd/_ab/lib_<name>.so.
# Usage: tools/ab_libs.sh name1:"-DIA_X=0 -DIA_Y=1" name2:"..." ; run with IA_LIB=imitation_amd/_ab/lib_<name>.so
set -e
cd "$(dirname "$0")/../imitation_amd/csrc"
mkdir -p ../_ab
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-function -Wno-unused-variable"
OTHERS="gemm.o mlp.o disc_fused.o airl_fused.o ppo_general.o host.o conv.o conv1_implicit.o conv3x3.o"
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  ( /opt/rocm/bin/hipcc $FLAGS $defs -c policy.hip -o ../_ab/policy_$name.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS ../_ab/policy_$name.o -o ../_ab/lib_$name.so &&
    rm -f ../_ab/policy_$name.o && echo "built lib_$name.so ($defs)" ) &
done
wait
