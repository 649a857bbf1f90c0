"""The tuning knobs only move work between the matcher's paths: the edge-case parity
tests must pass unchanged for every setting.  Each setting runs in its own process (the knobs are read once, at
context creation)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SETTINGS = [
    {"PGICP_KX": "1"},
    {"PGICP_MED_RINGS": "1", "PGICP_FAST_RINGS_SEEDED": "3", "PGICP_FAST_RINGS_UNSEEDED": "1"},
    {"PGICP_KX": "7", "PGICP_MED_RINGS": "8", "PGICP_POLL_US": "0", "PGICP_NEAR_FRAC": "0.05"},
    # round 2: the wave-per-query path without its bounded stage and with the old slack / ring settings ...
    {"PGICP_SLOW_SQUARE_ROWS": "0", "PGICP_MED_SHORT_RINGS": "4", "PGICP_PRUNE_PCT": "15", "PGICP_SEL_SMALL_N": "0"},
    # ... with a small bounded stage, no slack, few queries per wave in the fast kernel ...
    {"PGICP_SLOW_SQUARE_ROWS": "16", "PGICP_PRUNE_PCT": "0", "PGICP_FAST_LANES": "16", "PGICP_SLOW_BLOCKS": "64"},
    # ... and with every iteration replayed from a captured graph, the one-kernel selection for every size
    {"PGICP_GRAPH_MAX_P": "4096", "PGICP_SEL_SMALL_N": "100000000", "PGICP_FAST_LANES": "4"},
    # round 6: every map with the succinct cell table (MapDev::sw: occupancy words + per-occupied-cell starts) instead of the dense ones
    {"PGICP_TABLES": "succinct"},
    {"PGICP_TABLES": "succinct", "PGICP_KX": "1", "PGICP_FAST_LANES": "16"},
    # round 6: the outlier selection's path is chosen by batch size (band path from four problems on): either path forced for every size,
    # and the seeded overlap probe's fast pass with one ring
    {"PGICP_SEL_BAND": "0", "PGICP_PROBE_RINGS": "1"},
    {"PGICP_SEL_BAND": "1", "PGICP_SEL_SMALL_N": "0"},
    # ... without the next iteration's matcher pass enqueued ahead of the convergence flag, small results copied directly
    {"PGICP_SPECULATE": "0", "PGICP_D2H_DIRECT": "1", "PGICP_PROBE_CAP_SCALE": "0.5"},
]


@pytest.mark.parametrize("setting", SETTINGS, ids=lambda s: ",".join(f"{k[6:]}={v}" for k, v in s.items()))
def test_parity_holds_for_every_knob_setting(setting):
    env = dict(os.environ, **setting)
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_edge_cases.py", "tests/test_gpu_matcher_state.py", "tests/test_local_mapper.py", "tests/test_gpu_chain.py",
                        "tests/test_gpu_bit_exact.py", "tests/test_gpu_parity.py::test_seeded_partial_chain_equals_the_unseeded_one",
                        "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"], cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
