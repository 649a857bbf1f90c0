#!/bin/bash
# the 25-40 ms pauses of a young process: libpgicp or the platform?  (tools/micro/stall_probe.hip: eight tiny kernels + a polled host flag)
OUT=gpurun_out/${1:-sp}; mkdir -p $OUT
P=./tools/micro/stall_probe
{
echo "first GPU process on the box:";            $P 0
echo "2 s after it exited:";  sleep 2;           $P 0
echo "again, 2 s later:";     sleep 2;           $P 0
echo "this one allocates, touches and frees 40 GB first:"; sleep 2; $P 40
echo "2 s after THAT exited:"; sleep 2;          $P 0
echo "2 s later:"; sleep 2;                      $P 0
echo "20 s later:"; sleep 20;                    $P 0
echo "while holding 40 GB (untouched):"; sleep 2; $P 0 1500 40
echo "2 s after that:"; sleep 2;                 $P 0
} 2>&1 | tee $OUT/stall_probe.txt
