python3 bench.py --prepare-only >/dev/null 2>&1
R=$PWD
(cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/tr1 -o s -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-profile > $R/gpurun_out/tr1.log 2>&1)
python3 tools/trace_summary.py gpurun_out/tr1 | head -30
