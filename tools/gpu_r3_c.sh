#!/bin/bash
# round 3, call C: the from-files run after the set-up changes + the CLI tests
O=gpurun_out/r3c
mkdir -p $O
timeout 300 python tools/cli_e2e.py > $O/e2e.log 2>&1
cut -c1-900 $O/e2e.log
( time timeout 1200 python -m pytest tests/test_cli_gpu.py tests/test_reference_gpu.py -x -q -m gpu ) > $O/pytest_cli.log 2>&1
tail -5 $O/pytest_cli.log
