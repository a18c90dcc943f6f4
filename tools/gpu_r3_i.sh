#!/bin/bash
# round 3, call I: binary-trait Step 2 behind the ABI (score test + Firth / SPA corrections on the device): the ABI test against the oracle,
# then the C++ driver's binary / count trait cases against regenie's own output
O=gpurun_out/r3i
mkdir -p $O
( time timeout 900 python -m pytest tests/test_step2_bt_gpu.py -x -q -m gpu ) > $O/pytest_abi.log 2>&1
tail -15 $O/pytest_abi.log
( time timeout 1500 python -m pytest tests/test_cli_gpu.py -q -m gpu -k "step2" ) > $O/pytest_cli.log 2>&1
tail -15 $O/pytest_cli.log
