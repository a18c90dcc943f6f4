#!/bin/bash
# round 3, final evidence on the final build: PMC traffic (configs[1] and configs[2] in full), kernel statistics, the batch sequence,
# the default bench line (which then finds the traffic files of this build), the whole GPU test suite, smoke()
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=r3
O=gpurun_out/r3final
mkdir -p $O
C1="--steps 1 --warmup 0 --no-cpu --no-disk --no-extra"
C3="--samples 500000 --snps 500000 --phenos 10 --steps 1 --warmup 0 --no-cpu --no-disk --no-extra"
RG_PIPELINES=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python bench.py $C1 > $O/fetch.log 2>&1
RG_PIPELINES=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python bench.py $C1 > $O/write.log 2>&1
RG_PIPELINES=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma -- python bench.py $C1 > $O/mfma.log 2>&1
python tools/pmc_summary.py $O/fetch $O/${R}_pmc_fetch.md > /dev/null
python tools/pmc_summary.py $O/write $O/${R}_pmc_write.md > /dev/null
python tools/pmc_summary.py $O/mfma $O/${R}_pmc_mfma.md > /dev/null
NB=$(python -c "import csv,glob; f=glob.glob('$O/fetch/**/*counter_collection.csv',recursive=True)[0]; print(sum(1 for r in csv.DictReader(open(f)) if 'k_bed_prep_rows' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE'))")
python tools/pmc_traffic.py $O/fetch $O/write $NB $O/${R}_traffic.json 109 1 | cut -c1-300
cp $O/${R}_traffic.json profiles/
rm -rf $O/fetch $O/write $O/mfma
RX='k_l1_gram128|k_l1_wty|k_reduce_slices|k_sum_folds'      # restricted to the level-1 kernels: the unrestricted --pmc passes of this size crash inside rocprofv3
( RG_PIPELINES=1 timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --kernel-include-regex "$RX" --output-format csv -d $O/fetch3 -- python bench.py $C3 ) > $O/fetch3.log 2>&1
( RG_PIPELINES=1 timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --kernel-include-regex "$RX" --output-format csv -d $O/write3 -- python bench.py $C3 ) > $O/write3.log 2>&1
python tools/pmc_traffic.py $O/fetch3 $O/write3 10 $O/${R}_config3_traffic.json 512 10 | cut -c1-300
cp $O/${R}_config3_traffic.json profiles/
rm -rf $O/fetch3 $O/write3
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 4 --warmup 1 --no-cpu --no-disk --no-extra > $O/stats.log 2>&1
python tools/prof_summary.py $O/stats $O/${R}_kernel_stats.md > /dev/null
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${R}_kernel_stats.csv
RG_PIPELINES=1 rocprofv3 --kernel-trace --output-format csv -d $O/seq -- python bench.py $C1 > $O/seq.log 2>&1
python tools/trace_seq.py $O/seq $O/${R}_batch_sequence.md > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats3 -- python bench.py $C3 > $O/stats3.log 2>&1
python tools/prof_summary.py $O/stats3 $O/${R}_config3_kernel_stats.md > /dev/null
rm -rf $O/stats $O/seq $O/stats3
echo "stamp before the tests: $(cat regenie_amd/lib/build.stamp)"
( time timeout 1500 python -m pytest tests -q -m gpu ) > $O/pytest_gpu.log 2>&1
tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
echo "stamp after tests and smoke (build() must have found the library current): $(cat regenie_amd/lib/build.stamp)"
( time python bench.py ) > $O/bench.log 2>&1; grep '^{' $O/bench.log | tail -1 > $O/${R}_bench_line.json
python - <<PY
import json
d=json.load(open("$O/${R}_bench_line.json"))
print("bench", d["ms_per_step"], d["value"], "roofline", {k:d["roofline"].get(k) for k in ("frac","achieved","traffic")})
print("e2e", d["end_to_end_from_files"] and d["end_to_end_from_files"].get("walls_s"))
c=d.get("config3_single_gpu") or {}
print("config3", {k:c.get(k) for k in ("ms_per_step","value","error")}, (c.get("roofline") or {}).get("frac"), (c.get("roofline") or {}).get("traffic"))
s2=d.get("step2") or {}
print("step2", s2.get("error") or {k:(round(v["ms_per_block"],3), round(v["variants_per_s"]/1e6,3)) for k,v in s2["cases"].items()})
print("cpu", {k:d["cpu_baseline"].get(k) for k in ("value","unit","cores","kind")})
PY
head -14 $O/${R}_kernel_stats.md
