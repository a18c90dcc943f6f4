#!/bin/bash
# round 3, call AF: --step 2 --bt --firth --approx from files at 50,000 x 100,000 (corrections on the device), next to regenie itself
O=gpurun_out/r3af
mkdir -p $O
( time timeout 900 python tools/cli_e2e_step2.py 50000 100000 1000 1 ) > $O/step2_bt_e2e.log 2>&1
cut -c1-600 $O/step2_bt_e2e.log | tail -12
rm -rf /tmp/e2e
