#!/bin/bash
# Per-launch durations of the batched Cholesky in isolation (tools/chol_probe.py under rocprofv3 --kernel-trace):
# prints the kernel sequence of the last repetition (name:microseconds) and its total.
# Usage (on the GPU box): bash tools/probe_dbg.sh [n] [batch]      default 1024 1375 (one level-0 batch of config 2)
cd /tmp && export TMPDIR=/tmp && cd ${GRAFT_REPO_ROOT:-/root/repo}
N=${1:-1024}; B=${2:-1375}
O=gpurun_out/cp
rm -rf $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python tools/chol_probe.py $N $B 2 > $O.log 2>&1
grep "max rel" $O.log
python - <<PY
import csv,glob
f=glob.glob('$O/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sel=[r for r in rows if 'k_chol' in r['Kernel_Name']]
n=len(sel)//2
print(' '.join('%s:%.0f' % (r['Kernel_Name'].split('(')[0].replace('void ','').split('<')[0][7:12], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in sel[n:]))
print('total %.0f us' % sum((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in sel[n:]))
PY
rm -rf $O
