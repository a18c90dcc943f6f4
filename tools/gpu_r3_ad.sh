#!/bin/bash
# round 3, call AD: --t2e at 500,000 samples, driver against the oracle
O=gpurun_out/r3ad
mkdir -p $O
( time timeout 1500 python -m pytest tests/test_reference_gpu.py -x -q -m gpu -s -k "t2e_vs_oracle_at_500k" ) > $O/pytest.log 2>&1
tail -12 $O/pytest.log | cut -c1-300
