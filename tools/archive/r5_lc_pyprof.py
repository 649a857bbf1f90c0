#!/usr/bin/env python3
"""Host-side profile of the loop-closure step (bench.py --workload loopclosure): where the Python wrapper spends the time the GPU idles."""
import cProfile, pstats, sys, os, io
sys.argv = ["bench.py", "--workload", "loopclosure", "--pairs", "512", "--steps", "6", "--warmup", "2", "--no-cpu-baseline", "--no-profile"]
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
pr = cProfile.Profile()
pr.enable()
bench.main()
pr.disable()
s = io.StringIO()
st = pstats.Stats(pr, stream=s).sort_stats("cumulative")
st.print_callees("align_local|set_maps|align_batch|allgather_edges|step")
print(s.getvalue()[:12000])
