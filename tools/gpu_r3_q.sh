#!/bin/bash
# round 3, call Q: Cox ridge level 1 (rg_l1_cox) against the oracle
O=gpurun_out/r3q
mkdir -p $O
( time timeout 900 python -m pytest tests/test_l1_cox_gpu.py -x -q -m gpu ) > $O/pytest.log 2>&1
tail -40 $O/pytest.log | cut -c1-300
