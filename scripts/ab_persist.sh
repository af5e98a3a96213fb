#!/bin/bash
# A/B: one launch per alignment (NGICP_PERSIST=1, the persistent pass kernel) against one launch per pass (0): align / loop times of
# scripts/prof_c3.py for the given configs (default c3 c5)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for c in "${@:-c3 c5}"; do
  for p in 0 1; do
    echo "== $c NGICP_PERSIST=$p"
    NGICP_PERSIST=$p timeout -k 10 120 python3 scripts/prof_c3.py 6 $c 2>&1 | tail -3
  done
done
