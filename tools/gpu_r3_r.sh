#!/bin/bash
# round 3, call R: the driver's --t2e against regenie's own files; the reference / CLI suites after the driver changes
O=gpurun_out/r3r
mkdir -p $O
( time timeout 1200 python -m pytest tests/test_reference_gpu.py tests/test_l1_cox_gpu.py -x -q -m gpu ) > $O/pytest_ref.log 2>&1
grep "passed\|failed\|error" $O/pytest_ref.log | tail -3
grep -E "^E " $O/pytest_ref.log | head -20 | cut -c1-300
( time timeout 1200 python -m pytest tests/test_cli_gpu.py -x -q -m gpu ) > $O/pytest_cli.log 2>&1
grep "passed\|failed\|error" $O/pytest_cli.log | tail -3
