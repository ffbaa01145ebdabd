#!/bin/bash
# Diagnostic: alternate several builds of the library on one device.  usage: ab_libs.sh "<case> <dtype>" lib1.so lib2.so ...
args="$1"; shift
for round in 1 2 3; do
  for lib in "$@"; do DPLL_ABI=$([ "$(basename $lib)" = r1.so ] && echo 7 || echo 12) python tools/diag/time_lib.py "$lib" $args 2>&1 | grep -v amdgpu.ids; done
done
