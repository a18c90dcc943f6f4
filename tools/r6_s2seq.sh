#!/bin/bash
# kernel sequence of one masked hard-call block of the Step-2 record (rocprofv3 kernel trace): name, duration
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/s2seq_raw -- python tools/step2_record.py > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob('gpurun_out/s2seq_raw/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_s2_compact_rows' in r['Kernel_Name']]
i = idx[-1]
a = i
while a > 0 and 'k_s2_rows' not in rows[a]['Kernel_Name']: a -= 1
b = i
while b < len(rows) - 1 and 'k_s2_packed_final' not in rows[b]['Kernel_Name']: b += 1
t0 = int(rows[a]['Start_Timestamp'])
for r in rows[a:b + 1]:
    print('%-28s start %8.1f us  dur %8.1f us' % ((lambda nm: (nm[nm.find('k_'):] if 'k_' in nm else nm).split('(')[0].split('<')[0][:28])(r['Kernel_Name']), (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
PY
rm -rf gpurun_out/s2seq_raw
