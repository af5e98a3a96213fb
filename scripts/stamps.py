import sys, numpy as np
raw = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 24)
a = raw.astype(np.float64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
names = ["entry", "prologue loads", "query cells", "rows listed", "ring1 done", "far rows done", "shells done", "loop end", "reduce done", "barrier"]
print("waves", len(a), " kernel span (cycles):", a[:, :10].max() - t0)
print("wave start offset percentiles [0,50,90,100]:", np.percentile(a[:, 0] - t0, [0, 50, 90, 100]).round(0))
prev = a[:, 0]
for k in range(1, 10):
    cur = a[:, k]
    ok = cur > 0
    d = (cur - prev)[ok]
    if ok.sum():
        print(f"{names[k]:16s} n={ok.sum():5d} delta cycles p10/p50/p90/max: {np.percentile(d,10):9.0f} {np.percentile(d,50):9.0f} {np.percentile(d,90):9.0f} {d.max():9.0f}")
    prev = np.where(ok, cur, prev)
life = a[:, 9] - a[:, 0]
print("wave lifetime p10/p50/p90/max:", np.percentile(life, [10, 50, 90, 100]).round(0))

g = a[:, 10]; nl = a[:, 11]; qc = a[:, 12]; bb = a[:, 13].astype(np.int64)
print("grow histogram:", {int(k): int((g == k).sum()) for k in np.unique(g)})
print("live rows p10/p50/p90/max:", np.percentile(nl, [10, 50, 90, 100]))
print("queries per batch p10/p50/p90:", np.percentile(qc, [10, 50, 90]), "mean", qc.mean())
print("box dims x/y/z median:", np.median(bb & 255), np.median((bb >> 8) & 255), np.median((bb >> 16) & 255), "max", (bb & 255).max(), ((bb >> 8) & 255).max(), ((bb >> 16) & 255).max())
st = a[:, 3] - a[:, 2]
for k in np.unique(g):
    m = g == k
    print(f"  grow {int(k)}: n={m.sum()} staging cycles median {np.median(st[m]):.0f} p90 {np.percentile(st[m], 90):.0f}; live rows median {np.median(nl[m]):.0f}; far-phase median {np.median((a[:,5]-a[:,4])[m]):.0f}")
life = a[:, 8] - a[:, 0]
order = np.argsort(-life)[:12]
print("slowest waves: life | prologue, cells, stage, ring1, far, shells, tail, reduce | grow nlive qcount box")
for w in order:
    d = [a[w, k] - a[w, k - 1] if a[w, k] > 0 and a[w, k-1] > 0 else -1 for k in range(1, 9)]
    x14 = int(a[w, 14]); x15 = int(a[w, 15])
    print(int(life[w]), [int(x) for x in d], int(g[w]), int(nl[w]), int(qc[w]), (int(bb[w]) & 255, (int(bb[w]) >> 8) & 255, (int(bb[w]) >> 16) & 255),
          "max-lane ring1: steps", x15 >> 48, "rows", (x15 >> 32) & 0xffff, "cand", x15 & 0xffffffff, "| far: steps", x14 >> 48, "walked", (x14 >> 32) & 0xffff, "rows", (x14 >> 16) & 0xffff, "cand", x14 & 0xffff)
print("sum of wave lifetimes (to reduce-done):", life.sum(), " / 3072 slots =", life.sum() / 3072)

# ring 1 in detail: [3] rows listed -> [16] units queued (bounds round trip + pushes) -> [17] queue drained -> [4]
m = (a[:, 16] > 0) & (a[:, 17] > 0)
for name, d in (("ring1: bounds + enqueue", a[m, 16] - a[m, 3]), ("ring1: drain queue", a[m, 17] - a[m, 16]), ("ring1: final sync", a[m, 4] - a[m, 17])):
    print(f"{name:26s} p10/p50/p90/max: {np.percentile(d,10):9.0f} {np.percentile(d,50):9.0f} {np.percentile(d,90):9.0f} {d.max():9.0f}")
print("units per wave p50/p90/max:", np.percentile(a[m, 19], [50, 90, 100]), " max units popped by one lane p50/p90/max:", np.percentile(a[m, 18], [50, 90, 100]))
x15 = raw[raw[:, 0] > 0][m, 15]; steps = (x15 >> np.uint64(48)).astype(int)
drain = a[m, 17] - a[m, 16]
A = np.vstack([np.ones_like(steps), steps]).T; coef = np.linalg.lstsq(A, drain, rcond=None)[0]
print("drain ~ %.0f + %.0f * (max-lane window steps)" % tuple(coef), " corr", np.corrcoef(drain, steps)[0, 1])
