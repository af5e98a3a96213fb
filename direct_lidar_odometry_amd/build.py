"""Build the HIP extension in-tree:  python -m direct_lidar_odometry_amd.build

Produces direct_lidar_odometry_amd/libngicp_hip.so for gfx950 (hipcc cross-compiles without a GPU).
-ffp-contract=off: FP32 squared distances must not be fused (the reference builds without -march,
/root/reference/CMakeLists.txt:14-15, so its distances are un-fused mul/add).
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libngicp_hip.so")
SOURCES = ["ngicp_api.hip", "ngicp_filters.hip"]
HEADERS = ["ngicp_pass_group.inc", "ngicp_math.h", "ngicp_grid.h", "ngicp_knn.h", "ngicp_pass.h", "ngicp_pass_st.h", "ngicp_cloudops.h", "ngicp_filters.h", os.path.join("..", "..", "include", "ngicp.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wall", "-Wno-unused-function"]


def hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False, extra_flags: list[str] | None = None) -> str:
    if not force and not needs_build():
        return LIB
    tmp = f"{LIB}.{os.getpid()}.tmp"  # unique per process: several ranks may find the library stale at the same time
    cmd = [hipcc(), *FLAGS, *(extra_flags or []), *[os.path.join(CSRC, s) for s in SOURCES], "-o", tmp]
    if verbose:
        print(" ".join(cmd), flush=True)
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout + res.stderr)
        raise RuntimeError("hipcc failed")
    if verbose and res.stderr:
        sys.stderr.write(res.stderr)
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
