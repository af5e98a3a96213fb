#!/bin/bash
# average duration of the working launches of the two hot kernels (rocprofv3 kernel trace of scripts/prof_c3.py), per config and per
# tuning build under variants/ (or the default library when there is none)
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
LIBS=$(ls $ROOT/variants/libngicp_*.so 2>/dev/null); [ -z "$LIBS" ] && LIBS=default
for lib in $LIBS; do
for c in "${@:-c3 c5}"; do
  rm -rf /tmp/kt_$c
  if [ "$lib" = default ]; then unset NGICP_LIB; else export NGICP_LIB=$lib; fi
  timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$c -- python3 $ROOT/scripts/prof_c3.py 4 $c > /tmp/kt_$c.log 2>&1
  echo "== $(basename $lib) $c: $(grep ^align /tmp/kt_$c.log | tail -1)"; tail -3 /tmp/kt_$c.log
  python3 - <<PY
import csv,glob
f=glob.glob('/tmp/kt_$c/**/*kernel_trace.csv',recursive=True)[0]
d={}
for r in csv.DictReader(open(f)):
    d.setdefault(r['Kernel_Name'][:40],[]).append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
for k,v in d.items():
    if 'gicp_pass' in k or 'gicp_queue' in k:
        w=[x for x in v if x>15]; print('   pass', round(sum(w)/len(w),2), len(w))
        n=len(w)//4
        if n: print('   pass by position in the align (last align):', [round(x,1) for x in w[-n:]])
    if 'lm_solve' in k:
        w=[x for x in v if x>5.5]
        if w: print('   solve', round(sum(w)/len(w),2), len(w), 'min', min(v))
PY
done; done
