#!/bin/bash
# round 3, call U: HBM traffic of the level-1 Gram at BASELINE configs[2] in full -- the --pmc passes restricted to the level-1 kernels
# (the unrestricted passes of this size crashed inside rocprofv3)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r3u
mkdir -p $O
C3="--samples 500000 --snps 500000 --phenos 10 --steps 1 --warmup 0 --no-cpu --no-disk --no-extra"
RX='k_l1_gram128|k_l1_wty|k_reduce_slices|k_sum_folds'
( RG_PIPELINES=1 timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "$RX" --output-format csv -d $O/fetch3 -- python bench.py $C3 ) > $O/fetch3.log 2>&1
echo "fetch rc $?"
( RG_PIPELINES=1 timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "$RX" --output-format csv -d $O/write3 -- python bench.py $C3 ) > $O/write3.log 2>&1
echo "write rc $?"
python tools/pmc_traffic.py $O/fetch3 $O/write3 10 $O/r3_config3_traffic.json 512 10 | cut -c1-600
find $O -name "*.csv" -size +30M -delete; find $O -name "*.db" -delete
tail -3 $O/fetch3.log | cut -c1-200
