"""A short run of the randomised parity stress (tools/stress_parity.py): random scenes, poses ahead of / beside short maps,
chain parameters (maxDist, trim ratio, MedianDist factors), iteration counts, float and double -- matcher state against the
oracle in every case.  (Minutes of it were run while the round's kernels changed: 6 381 cases equal.)"""
import os, subprocess, sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_random_cases_equal_the_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_parity.py"), "25", "3"], cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "all equal to the oracle" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_random_chains_with_the_other_modules_equal_the_oracle():
    """tools/stress_chain.py: random batches with a random chain each (knn 1-4, PointToPoint, force4DOF, SurfaceNormal / MaxDist /
    MedianDist outlier filters, Bound checker loose and tight) -- whole ICP runs against the oracle"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stress_chain.py"), "20", "5"], cwd=ROOT, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "all equal to the oracle" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
