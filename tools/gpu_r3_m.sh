#!/bin/bash
# round 3, call M: BASELINE configs[2] FROM FILES on one GPU (a 62.5 GB .bed written to the box's disk), the null-Firth file test
O=gpurun_out/r3m
mkdir -p $O
( time timeout 300 python -m pytest tests/test_reference_gpu.py::test_driver_write_and_use_null_firth -q -m gpu ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log | cut -c1-300
df -h /tmp | tail -1
( time timeout 1500 python tools/cli_e2e.py 500000 500000 10 ) > $O/e2e_config3.log 2>&1
cut -c1-1200 $O/e2e_config3.log
rm -rf /tmp/e2e
