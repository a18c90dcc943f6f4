#!/bin/bash
# round 3, call L: step 2 on several contexts, null-Firth files
O=gpurun_out/r3l
mkdir -p $O
( time timeout 900 python -m pytest tests/test_cli_gpu.py -q -m gpu -k "step2_multi_gpu or usage" tests/test_reference_gpu.py::test_driver_write_and_use_null_firth ) > $O/pytest.log 2>&1
tail -25 $O/pytest.log | cut -c1-300
