import os, sys, subprocess
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
for vox in (0.0, 0.12, 0.15, 0.2, 0.25, 0.3):
    for grow in (6, 3, 2, 1):
        env = dict(os.environ, NGICP_STAGE_GROW=str(grow), NGICP_VOXEL=str(vox))
        r = subprocess.run([sys.executable, "scripts/prof_c3.py", "4", cfg], env=env, capture_output=True, text=True)
        lines = [l for l in r.stdout.splitlines() if l.startswith("align")]
        print(f"grow {grow} vox {vox}: {min(lines, key=lambda l: float(l.split()[4])) if lines else r.stderr[-300:]}", flush=True)
