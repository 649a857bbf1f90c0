timeout 900 python -m pytest tests/test_gpu_edge_cases.py -m gpu -x -q 2>&1 | tail -15
