cd $GRAFT_REPO_ROOT
for cfg in c3 c5; do for r in 0 10 30 45; do echo "== $cfg rot $r"; ROT_DEG=$r timeout -k 10 120 python scripts/prof_c3.py 4 $cfg 2>&1 | grep "^align" | tail -1; done; done
