"""BASELINE.md section 2's configure-time probe -- TEST INFRASTRUCTURE (only bench.py's cpu_baseline leg and tests use it).

Looks for an INSTALLED libpointmatcher (the library pgslam's ICP arithmetic lives in, un-vendored and un-pinned:
/root/reference/CMakeLists.txt:17-18) together with what its headers need (Eigen3, libnabo, Boost, yaml-cpp).  If all of it
is there, oracle/pm_ref_harness.cpp is compiled into oracle/_ref/pm_ref and `run()` times the real library on the
benchmark's problems: cpu_baseline.kind "reference", and the golden fixtures can be pinned through it
(tests/test_oracle.py::test_golden_fixtures_through_installed_libpointmatcher).  If anything is missing -- the case in
this image and on the GPU boxes of this pool -- `probe()` says what was looked for and the bench keeps the port.
Nothing is downloaded, nothing is stubbed."""
import ctypes.util
import glob
import json
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_DIR = os.path.join(HERE, "_ref")


def _prefixes():
    out = []
    for var in ("PGSLAM_LIBPOINTMATCHER_PREFIX", "CMAKE_PREFIX_PATH", "CONDA_PREFIX"):
        for p in os.environ.get(var, "").split(os.pathsep):
            if p:
                out.append(p)
    return out + ["/usr/local", "/usr", "/opt/local", "/opt/ros/noetic", "/opt/ros/humble"]


def _find_header(rel):
    for p in _prefixes():
        for inc in ("include", "include/eigen3"):
            c = os.path.join(p, inc, rel)
            if os.path.exists(c):
                return os.path.join(p, inc)
    return None


def _find_lib(name):
    hit = ctypes.util.find_library(name)
    if hit:
        return hit
    for p in _prefixes():
        for sub in ("lib", "lib64", "lib/x86_64-linux-gnu"):
            g = glob.glob(os.path.join(p, sub, f"lib{name}.so*")) + glob.glob(os.path.join(p, sub, f"lib{name}.a"))
            if g:
                return g[0]
    return None


def probe(build=True):
    """-> dict(found, missing, include_dirs, exe | None, build_error | None)"""
    need_h = {"libpointmatcher": "pointmatcher/PointMatcher.h", "Eigen3": "Eigen/Core", "libnabo": "nabo/nabo.h",
              "Boost": "boost/version.hpp", "yaml-cpp": "yaml-cpp/yaml.h"}
    inc, missing = {}, []
    for k, h in need_h.items():
        d = _find_header(h)
        if d:
            inc[k] = d
        else:
            missing.append(f"{k} header {h}")
    libs = {}
    for k in ("pointmatcher", "nabo", "yaml-cpp"):
        p = _find_lib(k)
        if p:
            libs[k] = p
        else:
            missing.append(f"lib{k}")
    out = dict(found=not missing, missing=missing, include_dirs=sorted(set(inc.values())), libraries=libs, exe=None, build_error=None,
               looked_in=_prefixes())
    if missing or not build:
        return out
    os.makedirs(REF_DIR, exist_ok=True)
    exe = os.path.join(REF_DIR, "pm_ref")
    cmd = ["g++", "-std=c++17", "-O2", os.path.join(HERE, "pm_ref_harness.cpp"), "-o", exe] + [f"-I{d}" for d in out["include_dirs"]]
    for p in libs.values():
        cmd += [f"-L{os.path.dirname(p)}"] if os.path.isabs(p) else []
    cmd += ["-lpointmatcher", "-lnabo", "-lyaml-cpp", "-lboost_system", "-lboost_filesystem", "-lboost_thread", "-lboost_chrono",
            "-lboost_timer", "-pthread"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        out["build_error"] = r.stderr[-2000:]
        return out
    out["exe"] = exe
    return out


def run(exe, reading, ref_xyz, ref_nrm, T_init, repetitions=1):
    """One problem through the installed libpointmatcher (oracle/_ref/pm_ref): dict(T, overlap, seconds, cov)."""
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        f.write(np.array([reading.shape[0], ref_xyz.shape[0]], dtype=np.int32).tobytes())
        f.write(np.ascontiguousarray(T_init, dtype=np.float64).tobytes())
        for a in (reading, ref_xyz, ref_nrm):
            f.write(np.ascontiguousarray(a[:, :3], dtype=np.float32).tobytes())
        path = f.name
    try:
        r = subprocess.run([exe, path, str(repetitions)], capture_output=True, text=True, check=True)
    finally:
        os.unlink(path)
    d = json.loads(r.stdout.strip().splitlines()[-1])
    d["T"] = np.array(d["T"]).reshape(4, 4)
    d["cov"] = np.array(d["cov"]).reshape(6, 6)
    return d


if __name__ == "__main__":
    print(json.dumps({k: v for k, v in probe().items()}, indent=1))
