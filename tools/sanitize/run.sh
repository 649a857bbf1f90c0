#!/bin/bash
# Sanitizer runs of the host-only code (CPU box; GPU AddressSanitizer is not available on the pool):
#   1. address + undefined and 2. thread sanitizer over
#      - the C++ drop-in's CPU tests (header-only code: pointmatcher.hpp, pgslam.hpp, slam.hpp -- graph, optimizer, queues, MT workers)
#      - the host transport of pgicp_allgather_edges, pgicp_comm.cpp compiled with the sanitizer, ranks as threads
# The product library itself (libpgicp.so) is linked uninstrumented for the symbols the tests never reach without a GPU.
# Output: tools/sanitize/out/*.log; a summary line per run on stdout.
cd "$(dirname "$0")/../.."
OUT=tools/sanitize/out; mkdir -p $OUT; rm -f $OUT/*.log
INC="-Iinclude -Ipgslam_amd/csrc -I/opt/rocm/include -D__HIP_PLATFORM_AMD__"
LINK="-Lpgslam_amd/lib -lpgicp -Wl,-rpath,$PWD/pgslam_amd/lib -Wl,-rpath,/opt/rocm/lib -L/opt/rocm/lib -lamdhip64 -ldl -pthread"
run() { # tag, sanitizer flags, env, exe args...
  local tag=$1; shift; local log=$OUT/$tag.log
  ( "$@" ) > $log 2>&1; local rc=$?
  local findings=$(grep -c "ERROR: \|WARNING: ThreadSanitizer\|runtime error:" $log)
  echo "$tag rc=$rc sanitizer_reports=$findings"
}
for SAN in asan tsan; do
  if [ $SAN = asan ]; then F="-fsanitize=address,undefined -fno-omit-frame-pointer"; export ASAN_OPTIONS=detect_leaks=1:halt_on_error=0 UBSAN_OPTIONS=print_stacktrace=1
  else F="-fsanitize=thread"; export PGSLAM_TEST_TIME_SCALE=20 TSAN_OPTIONS="halt_on_error=0 second_deadlock_stack=1"; fi
  for t in test_dropin_cpu test_slam_cpu test_instantiation; do
    g++ -std=c++17 -O1 -g $F -pthread -Iinclude tests/cpp/$t.cpp -o $OUT/${t}_$SAN $LINK 2> $OUT/build_${t}_$SAN.log || { echo "build ${t}_$SAN FAILED"; continue; }
    HIP_VISIBLE_DEVICES=-1 ROCR_VISIBLE_DEVICES=-1 run ${t}_$SAN $OUT/${t}_$SAN
  done
  g++ -std=c++17 -O1 -g $F -pthread $INC tools/sanitize/comm_threads.cpp pgslam_amd/csrc/pgicp_comm.cpp -o $OUT/comm_threads_$SAN $LINK 2> $OUT/build_comm_$SAN.log || { echo "build comm_$SAN FAILED"; continue; }
  for w in 2 3 8; do rm -f /dev/shm/pgicp_sanitize_comm_$w; run comm_threads_w${w}_$SAN $OUT/comm_threads_$SAN $w 3 /dev/shm/pgicp_sanitize_comm_$w; done
done
rm -f $OUT/*_asan $OUT/*_tsan
