#!/bin/bash
# round 3, call Z: host rows cross PCIe as one linear copy per block (was a pitched copy); the step-1 suites, then the run from files at
# configs[1] and configs[2]
O=gpurun_out/r3z
mkdir -p $O
( time timeout 900 python -m pytest tests/test_step1_gpu.py tests/test_cli_gpu.py tests/test_l0_f64_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log | cut -c1-200
( time timeout 600 python tools/cli_e2e.py ) > $O/e2e_config2.log 2>&1
cut -c1-900 $O/e2e_config2.log | tail -8
( time timeout 1500 python tools/cli_e2e.py 500000 500000 10 ) > $O/e2e_config3.log 2>&1
cut -c1-1000 $O/e2e_config3.log | tail -6
rm -rf /tmp/e2e
