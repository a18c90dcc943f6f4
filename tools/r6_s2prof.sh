#!/bin/bash
# Step 2 (QT) record under rocprofv3: the cases' rates and the kernels behind them
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s2prof_raw -- python tools/step2_record.py > gpurun_out/s2rec.json 2>/dev/null
python tools/prof_summary.py gpurun_out/s2prof_raw gpurun_out/s2prof.md > /dev/null; rm -rf gpurun_out/s2prof_raw
grep "compact\|k_xy_i8\|k_s2_rows\|combine\|masked" gpurun_out/s2prof.md
python -c "
import json
d=json.load(open('gpurun_out/s2rec.json'))
for k,v in d['cases'].items(): print(k, v['ms_per_block'], v['variants_per_s'])
"
