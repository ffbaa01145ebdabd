#!/bin/bash
# Diagnostic: lane-per-contact builds (DPLL_WIDE=0) against the one-lane-per-item build (DPLL_WIDE=1) over batch sizes.
for b in 4096 16384 32768 65536 262144; do for w in 0 1; do echo -n "wide=$w B=$b "; DPLL_WIDE=$w python3 bench.py --no-cpu-baseline --batch $b --steps 300 --warmup 100 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']/1e6,1), 'M/s', round(d['ms_per_step']*1e3,1), 'us')"; done; done
