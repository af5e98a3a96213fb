import sys, numpy as np
for f in sys.argv[1:]:
    a = np.fromfile(f, dtype=np.int32); n = len(a)//3
    c, c2, sp = a[:n]*16/2400.0, a[n:2*n]*16/2400.0, a[2*n:]
    print(f, "groups", n, "split", sp.sum())
    un = c[sp == 0]
    print("  unsplit us p10/50/90/99/max:", np.percentile(un, [10,50,90,99,100]).round(1), "sum/768 slots:", round((un.sum() + c[sp==1].sum() + c2[sp==1].sum())/768,1))
    if sp.sum():
        print("  split halves A us p50/max:", np.percentile(c[sp==1],[50,100]).round(1), "B:", np.percentile(c2[sp==1],[50,100]).round(1))
