"""Builds and runs the C++ drop-in tests (g++ against include/ and libpgicp.so)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPP = os.path.join(ROOT, "tests", "cpp")


def build(name):
    exe = os.path.join(CPP, name)
    src = exe + ".cpp"
    # always rebuilt: a snapshot pushed to another box flattens mtimes, and a stale binary must never be what runs
    if True:
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-Wno-unused-local-typedefs", "-Wno-unused-variable", "-pthread", "-I" + os.path.join(ROOT, "include"), src, "-o", exe,
                               "-L" + os.path.join(ROOT, "pgslam_amd", "lib"), "-lpgicp",
                               "-Wl,-rpath," + os.path.join(ROOT, "pgslam_amd", "lib"), "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def test_dropin_cpu():
    out = subprocess.run([build("test_dropin_cpu")], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "dropin cpu tests ok" in out.stdout


def test_four_way_instantiation_without_a_device():
    """float / double x single / multi thread through the forwarding headers pgslam user code includes
    (reference tests/instantiation.cpp:4-19), the 3-string constructors and SetIcpConfig(paths) on both flavours."""
    out = subprocess.run([build("test_instantiation")], capture_output=True, text=True, timeout=120,
                         env=dict(os.environ, HIP_VISIBLE_DEVICES="-1", ROCR_VISIBLE_DEVICES="-1"))
    assert out.returncode == 0, out.stdout + out.stderr
    assert "instantiation tests ok" in out.stdout


@pytest.mark.gpu
def test_dropin_gpu():
    out = subprocess.run([build("test_dropin_gpu")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "dropin gpu tests ok" in out.stdout
