#!/bin/bash
# Round 6: the panel-128 Cholesky on the device -- parity tests that go through it, then the bench line, the kernel sequence of a batch and the
# kernel statistics, for the new path and (A/B) for the group-of-four kernels (RG_CHOL_GROUP4=1).
cd $GRAFT_REPO_ROOT
O=gpurun_out/r6_chol; mkdir -p $O
( time timeout 900 python -m pytest tests/test_kernels_gpu.py tests/test_step1_gpu.py -q -m gpu -x ) > $O/pytest.log 2>&1; tail -6 $O/pytest.log
OUT=r6_chol/bench bash tools/gpu_job.sh bench --steps 10 --warmup 2 --no-extra --no-disk --cpu-blocks 2 | head -12
OUT=r6_chol/seq bash tools/gpu_job.sh seq --steps 2 --warmup 1 --no-cpu --no-extra --no-disk | head -40
if [ -n "$AB" ]; then
RG_CHOL_GROUP4=1 OUT=r6_chol/bench_g4 bash tools/gpu_job.sh bench --steps 10 --warmup 2 --no-extra --no-disk --no-cpu | head -6
fi
