#!/bin/bash
# round 3, call AA: the mapped .bed as the copy source on every rank of a multi-GPU run; CLI + reference suites
O=gpurun_out/r3aa
mkdir -p $O
( time timeout 1500 python -m pytest tests/test_cli_gpu.py tests/test_reference_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2
grep -E "^E " $O/pytest.log | head -10 | cut -c1-300
