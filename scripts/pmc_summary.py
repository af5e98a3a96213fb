"""gpurun_out/pmc_<cfg>/* (scripts/pmc_profile.sh) -> profiles/<tag>_<cfg>_kernel_stats.csv + profiles/<tag>_<cfg>_pass_counters.json.
usage: python scripts/pmc_summary.py <cfg> [round tag, default r03]"""
import collections, csv, glob, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = sys.argv[1] if len(sys.argv) > 1 else "c3"
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
SRC = os.path.join(ROOT, "gpurun_out", f"pmc_{cfg}")
DST = os.path.join(ROOT, "profiles")
os.makedirs(DST, exist_ok=True)


def one(pattern):
    fs = glob.glob(os.path.join(SRC, pattern), recursive=True)  # (gpurun merges into what is already there: earlier runs' files stay)
    return max(fs, key=os.path.getmtime) if fs else None


stats = one("trace/**/*kernel_stats.csv")
rows = list(csv.DictReader(open(stats)))
with open(os.path.join(DST, f"{tag}_{cfg}_kernel_stats.csv"), "w") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
    for r in rows:
        w.writerow([r["Name"], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"], r["StdDev"]])
trace = one("trace/**/*kernel_trace.csv")
durs = collections.defaultdict(list)
for r in csv.DictReader(open(trace)):
    durs[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
pass_name = next(k for k in durs if "k_gicp_pass" in k)
solve_name = next(k for k in durs if "k_lm_solve" in k)
work = [d for d in durs[pass_name] if d > 15.0]   # launches after `done` exit immediately (~4 us)
swork = [d for d in durs[solve_name] if d > 3.0]
cov = {k: sum(v) / len(v) for k, v in durs.items() if "k_covariances" in k}


def pmc(name):
    """Counter rows of one PMC pass, restricted to the WORKING launches of each kernel: every pass carries its own kernel trace, so
    a counter row is joined to its dispatch's duration by dispatch id and kept when the launch ran longer than 15 us (the launches
    enqueued after the alignment has finished return at once).  (Round 2 dropped the smallest third of the values instead, which
    also threw genuine launches away and biased the means upward.)"""
    f = one(f"{name}/**/*counter_collection.csv")
    t = one(f"{name}/**/*kernel_trace.csv")
    dur = {}
    if t:
        for r in csv.DictReader(open(t)):
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            d = dur.get(r.get("Dispatch_Id"))
            if d is not None and d <= 15.0 and "k_gicp_pass" in r["Kernel_Name"]:
                continue
            agg[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def work_mean(agg, kern_sub, counter):
    for k, v in agg.items():
        if kern_sub in k and counter in v:
            return sum(v[counter]) / len(v[counter])
    return None


counters = {}
for name in ("sq1", "sq2", "ta1", "ta2", "tcp1", "tcp2", "tcp3", "tcp4", "td1", "l2", "fetch", "write", "grbm"):
    agg = pmc(name)
    for k, v in agg.items():
        if "k_gicp_pass" in k:
            for c in v:
                counters[c] = work_mean(agg, "k_gicp_pass", c)
fetch, write = counters.get("FETCH_SIZE"), counters.get("WRITE_SIZE")
summary = {
    "command": f"scripts/pmc_profile.sh {cfg}: rocprofv3 --kernel-trace --stats -- python3 scripts/prof_c3.py 3 {cfg}  (+ separate --pmc passes)",
    "pass_kernel": pass_name, "pass_launches_working": len(work), "pass_avg_us_working": sum(work) / len(work), "pass_min_us": min(work), "pass_max_us": max(work),
    "solve_avg_us_working": sum(swork) / max(1, len(swork)), "solve_launches_working": len(swork), "covariance_kernels_avg_us": cov,
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read on gfx950 -> doubled; WRITE_SIZE exact for
    # 16 B/lane stores.  The pass's reads are mostly 16-byte gathers (not a calibrated pattern): treat as an estimate of fabric traffic.
    "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0 if fetch is not None and write is not None else None,
    "counters_per_working_launch": counters,
}
json.dump(summary, open(os.path.join(DST, f"{tag}_{cfg}_pass_counters.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
