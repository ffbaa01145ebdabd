#!/bin/bash
# Diagnostic: builds of the general translation unit whose first N automatic variables are filled with clang's poison pattern
# (floats: NaN) -- tools/diag/poison_bisect.sh N [N ...] -> tools/diag/variants/poison_N.so.  A kernel that reads a local before
# writing it returns NaN / garbage deterministically from the N of that variable on.
cd "$(dirname "$0")/../../dair_pll_amd/csrc"
out=../../tools/diag/variants; mkdir -p $out
for N in "$@"; do
  ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -fno-slp-vectorize -Wno-unused-function -Wno-unused-variable -Wno-pass-failed \
      -mllvm -amdgpu-sched-strategy=max-ilp -ftrivial-auto-var-init=pattern -ftrivial-auto-var-init-stop-after=$N -c -o $out/poison_$N.o dpll_general.hip > $out/poison_$N.log 2>&1 \
    && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/poison_$N.so dpll_kernels.o $out/poison_$N.o dpll_genmesh.o dpll_forest.o && rm -f $out/poison_$N.o ) &
done
wait
ls -la $out/poison_*.so
