#!/bin/bash
# randomised parity stress on the round's last kernels (bounce buffer, selection hints, rings by cell size): GPU against the CPU oracle
OUT=gpurun_out/${1:-stress}; mkdir -p $OUT
{
python tools/stress_chain.py 300 51 2>&1 | tail -2
python tools/stress_batch.py 200 52 2>&1 | tail -2
python tools/stress_parity.py 200 53 2>&1 | tail -2
python tools/stress_full_size.py 360 54 2>&1 | tail -2
} | grep -vE "^(RCCL|HIP version|ROCm version|Hostname|Librccl)" | tee $OUT/stress.txt
