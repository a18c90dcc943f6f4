cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for d in ${PROBE_SET:-0}; do
  export RG_CHOL_DBG=$d
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cp_${RG_CHOL_NS}_$d -- python tools/chol_probe.py 1024 1375 2 > gpurun_out/cp_${RG_CHOL_NS}_$d.log 2>&1
  echo "dbg=$d NS=$RG_CHOL_NS"; grep "max rel" gpurun_out/cp_${RG_CHOL_NS}_$d.log
  python - <<PY
import csv,glob
f=glob.glob('gpurun_out/cp_${RG_CHOL_NS}_$d/**/*kernel_trace.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
sel=[r for r in rows if 'k_chol' in r['Kernel_Name']]
n=len(sel)//2
print(' '.join('%s:%.0f' % (r['Kernel_Name'].split('(')[0].replace('void ','')[7:12], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in sel[n:]))
tot=sum((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in sel[n:]); print('total %.0f us' % tot)
PY
done
