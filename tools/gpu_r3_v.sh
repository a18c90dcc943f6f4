#!/bin/bash
# round 3, call V: the driver's --t2e on the second fixture (example genotypes, --remove / --cv 3 / --ref-first, time columns out of order)
O=gpurun_out/r3v
mkdir -p $O
( time timeout 900 python -m pytest tests/test_reference_gpu.py -x -q -m gpu -k "t2e" ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log | cut -c1-200
grep -E "^E " $O/pytest.log | head -10 | cut -c1-300
