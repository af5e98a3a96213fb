"""Tuning builds of libngicp_hip.so (loaded through NGICP_LIB): python scripts/build_variants.py name=flag[,flag] ..."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from direct_lidar_odometry_amd import build as b
os.makedirs(os.path.join(ROOT, "variants"), exist_ok=True)
procs = []
for spec in sys.argv[1:]:
    name, flags = spec.split("=", 1)
    out = os.path.join(ROOT, "variants", f"libngicp_{name}.so")
    cmd = [b.hipcc(), *b.FLAGS, *[f for f in flags.split(",") if f], *[os.path.join(b.CSRC, src) for src in b.SOURCES], "-o", out]
    procs.append((name, subprocess.Popen(cmd)))
for name, p in procs:
    print(name, "rc", p.wait())
