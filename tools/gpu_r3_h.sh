#!/bin/bash
# round 3, call H: the complete -m gpu suite on the current build
O=gpurun_out/r3h
mkdir -p $O
( time timeout 1700 python -m pytest tests -q -m gpu -x ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
