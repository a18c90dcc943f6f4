#!/bin/bash
# round-2 GPU session K: complete -m gpu suite on the final build + refreshed round-2 profiles
export TMPDIR=/tmp
O=gpurun_out/r2k
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $O/pytest.log 2>&1
ROUND=r2 timeout 900 bash tools/collect_profiles.sh > $O/collect.log 2>&1
