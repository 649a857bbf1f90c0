#!/bin/bash
python3 bench.py --prepare-only > /dev/null 2>&1
tools/ab_headline.sh "X=1" "PGICP_FAST_RINGS_UNSEEDED=2" "PGICP_FAST_RINGS_UNSEEDED=4" "PGICP_FAST_RINGS_UNSEEDED=1" "PGICP_MED_RINGS=6" "PGICP_MED_RINGS=2" "PGICP_KX=8" "PGICP_FAST_RINGS_SEEDED=2" "X=1" 2>&1 | tee gpurun_out/r3/ab_knobs.txt
