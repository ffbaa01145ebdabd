#!/bin/bash
# Diagnostic: one variant of the general translation unit linked against the shipped dpll_kernels.o.
#   tools/diag/build_variant.sh <name> "<flags replacing GENERAL_EXTRA>" ["<flags replacing KERNELS_SCHED>"]
# -> tools/diag/variants/<name>.so (git-ignored, travels to the GPU box); check with DPLL_LIB=... tools/diag/check_general.py
set -e
src=${DPLL_SRC:-$(dirname "$0")/../../dair_pll_amd/csrc}; out=$(cd "$(dirname "$0")" && pwd)/variants; cd "$src"
name=$1; extra=$2; sched=${3--mllvm -amdgpu-sched-strategy=max-ilp}

/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize -Wno-unused-function -Wno-unused-variable -Wno-pass-failed $sched $extra -c -o $out/$name.o dpll_general.hip
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/$name.so dpll_kernels.o $out/$name.o $( [ -f dpll_genmesh.o ] && echo dpll_genmesh.o )
rm -f $out/$name.o
