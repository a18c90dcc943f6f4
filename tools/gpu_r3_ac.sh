#!/bin/bash
# round 3, call AC: the driver leaves through _exit after a successful run (no buffer-by-buffer teardown): CLI suites, then the run from files
O=gpurun_out/r3ac
mkdir -p $O
( time timeout 1500 python -m pytest tests/test_cli_gpu.py tests/test_reference_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1
grep -E "passed|failed" $O/pytest.log | tail -2
( time timeout 600 python tools/cli_e2e.py ) > $O/e2e_config2.log 2>&1
cut -c1-330 $O/e2e_config2.log | tail -9
( time RG_TEARDOWN=1 timeout 600 python tools/cli_e2e.py ) > $O/e2e_config2_teardown.log 2>&1
cut -c1-330 $O/e2e_config2_teardown.log | tail -9
rm -rf /tmp/e2e
