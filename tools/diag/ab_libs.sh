#!/bin/bash
# Diagnostic: alternate several builds of the library on one device.  usage: ab_libs.sh "<case> <dtype>" lib1.so lib2.so ...
# Only builds of the CURRENT ABI can be compared: the ctypes struct layouts of dair_pll_amd/_capi.py belong to it, and the
# binding refuses a library whose dpll_abi_version() differs (no override).
args="$1"; shift
for round in 1 2 3; do
  for lib in "$@"; do python tools/diag/time_lib.py "$lib" $args 2>&1 | grep -v amdgpu.ids; done
done
