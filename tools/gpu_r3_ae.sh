#!/bin/bash
# round 3, call AE: configs[2] from files on the final build; the whole GPU suite once more (it gained the 500,000-sample --t2e case)
O=gpurun_out/r3ae
mkdir -p $O
( time timeout 1500 python tools/cli_e2e.py 500000 500000 10 ) > $O/e2e_config3.log 2>&1
cut -c1-700 $O/e2e_config3.log | sed -n 2,5p
rm -rf /tmp/e2e
( time timeout 1800 python -m pytest tests -q -m gpu ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log | head -1
