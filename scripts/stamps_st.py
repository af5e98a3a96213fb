"""Per-wave s_memtime stamps of the last pass (NGICP_DEBUG_STAMPS=<file>) of the staged pass kernel: where a wave's cycles go.
usage: python scripts/stamps_st.py <file>"""
import sys, numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 24)
keep = raw[:, 0] > 0
raw = raw[keep]
a = raw.astype(np.float64)
names = ["entry", "batch record", "operands + query setup", "round-0 rows + bounds", "round 0 done", "rounds done", "shells done", "tail done", "reduce done", "barrier"]
print("waves", len(a))
prev = a[:, 0]
for k in range(1, 10):
    cur = a[:, k]
    ok = cur > 0
    d = (cur - prev)[ok]
    if ok.sum():
        print(f"{names[k]:26s} n={ok.sum():5d} delta cycles p10/p50/p90/max: {np.percentile(d,10):9.0f} {np.percentile(d,50):9.0f} {np.percentile(d,90):9.0f} {d.max():9.0f}")
    prev = np.where(ok, cur, prev)
life = a[:, 8] - a[:, 0]
print("wave lifetime (to reduce done) p10/p50/p90/max:", np.percentile(life, [10, 50, 90, 100]).round(0), " sum / 1e6:", life.sum() / 1e6)
names2 = {10: "points staged", 11: "rows listed", 12: "chunks", 13: "units", 14: "drain passes", 15: "window steps of the busiest lane", 16: "search probes of the busiest lane", 17: "region rows", 18: "queries"}
for k in range(10, 19):
    v = a[:, k]
    if v.max() > 0:
        print(f"{names2[k]:34s} p10/p50/p90/max/mean: {np.percentile(v,10):.0f} {np.percentile(v,50):.0f} {np.percentile(v,90):.0f} {v.max():.0f} {v.mean():.1f}")

names2 = {10: "points staged", 11: "rows listed", 12: "chunks", 13: "units", 14: "drain passes", 15: "window steps of the busiest lane", 16: "search probes of the busiest lane", 17: "region rows", 18: "queries"}
r0 = a[:, 4] - a[:, 3]
order = np.argsort(-r0)[:10]
print("slowest round 0: cycles | " + ", ".join(names2[k] for k in range(10, 19)))
for w in order:
    print(int(r0[w]), [int(a[w, k]) for k in range(10, 19)])
X = np.vstack([np.ones(len(a)), a[:, 12], a[:, 13], a[:, 15], a[:, 16], a[:, 10]]).T
coef = np.linalg.lstsq(X, r0, rcond=None)[0]
print("round 0 ~ %.0f + %.0f chunks + %.1f units + %.0f window steps + %.0f probes + %.1f points" % tuple(coef))

if raw[:, 19].max() > 0:  # diagnostic build (-DNGICP_ST_DIAG): cycles per phase of the chunk pipeline, summed over a wave's chunks
    ph = {"copy issue": a[:, 19], "copy wait": a[:, 20], "unit list": a[:, 21], "A search + 8 around": a[:, 22],
          "B count blocks": (raw[:, 23] & np.uint64(0xffffffff)).astype(np.float64), "C sub-units": (raw[:, 23] >> np.uint64(32)).astype(np.float64)}
    for k, v in ph.items():
        print(f"{k:22s} p10/p50/p90/max/mean: {np.percentile(v,10):8.0f} {np.percentile(v,50):8.0f} {np.percentile(v,90):8.0f} {v.max():8.0f} {v.mean():8.0f}")
