#!/bin/bash
# round 3, call K: null-Firth files, 50 binary traits, rank-failure, then the whole -m gpu suite
O=gpurun_out/r3k
mkdir -p $O
( time timeout 900 python -m pytest tests/test_reference_gpu.py::test_driver_write_and_use_null_firth tests/test_l1_models_gpu.py::test_bt_kfold_fifty_phenotypes -x -q -m gpu ) > $O/pytest_new.log 2>&1
tail -12 $O/pytest_new.log | cut -c1-300
( time timeout 1700 python -m pytest tests -q -m gpu ) > $O/pytest_all.log 2>&1
tail -8 $O/pytest_all.log | cut -c1-300
