#!/bin/bash
# round-2 GPU sessions E..: the exact i8 routes (pred_i8.hip, xy_i8.hip): parity tests, config-3 per-GPU share, kernel stats
export TMPDIR=/tmp
R=${RUN:-r2h}
mkdir -p gpurun_out/$R
O=gpurun_out/$R
( time timeout 900 python -m pytest tests/test_step1_gpu.py tests/test_reference_gpu.py tests/test_distributed_gpu.py tests/test_fullsize_gpu.py tests/test_loocv_gpu.py tests/test_l1_models_gpu.py -m gpu -q -x 2>&1 | tail -40 ) > $O/pytest.log 2>&1
timeout 600 python bench.py --samples 500000 --snps 62500 --phenos 10 --no-cpu --steps 3 --warmup 1 > $O/config3_share.json 2> $O/config3_share.err
RG_XY_F64=1 timeout 600 python bench.py --samples 500000 --snps 62500 --phenos 10 --no-cpu --steps 3 --warmup 1 > $O/config3_share_xyf64.json 2>> $O/config3_share.err
timeout 300 python bench.py --no-cpu --steps 10 > $O/bench_nocpu.json 2>> $O/config3_share.err
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_e -- python $GRAFT_REPO_ROOT/bench.py --samples 500000 --snps 62500 --phenos 10 --no-cpu --steps 2 --warmup 1 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python tools/prof_summary.py /tmp/prof_e $O/r2_config3_share_kernel_stats.md > /dev/null 2>&1
