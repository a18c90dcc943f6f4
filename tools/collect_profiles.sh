#!/bin/bash
# Refreshes the per-round evidence under gpurun_out/ (copy the .md/.json files into profiles/ afterwards):
#   bench line, rocprofv3 kernel stats of the same command, the kernel sequence of one level-0 batch, and three PMC
#   passes (FETCH_SIZE, WRITE_SIZE, SQ/GRBM) -- counters in their own runs, never combined with API traces.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${ROUND:-r1}
O=gpurun_out/$R
mkdir -p $O
python bench.py > $O/bench.log 2>&1; tail -1 $O/bench.log > $O/${R}_bench_line.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python bench.py --steps 4 --warmup 1 --no-cpu > $O/stats.log 2>&1
python tools/prof_summary.py $O/stats $O/${R}_kernel_stats.md > /dev/null
cp $(find $O/stats -name "*kernel_stats.csv" | head -1) $O/${R}_kernel_stats.csv
RG_PIPELINES=1 rocprofv3 --kernel-trace --output-format csv -d $O/seq -- python bench.py --steps 1 --warmup 0 --no-cpu > $O/seq.log 2>&1
python tools/trace_seq.py $O/seq $O/${R}_batch_sequence.md > /dev/null
RG_PIPELINES=1 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch -- python bench.py --steps 1 --warmup 0 --no-cpu > $O/fetch.log 2>&1
RG_PIPELINES=1 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write -- python bench.py --steps 1 --warmup 0 --no-cpu > $O/write.log 2>&1
RG_PIPELINES=1 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/mfma -- python bench.py --steps 1 --warmup 0 --no-cpu > $O/mfma.log 2>&1
python tools/pmc_summary.py $O/fetch $O/${R}_pmc_fetch.md > /dev/null
python tools/pmc_summary.py $O/write $O/${R}_pmc_write.md > /dev/null
python tools/pmc_summary.py $O/mfma $O/${R}_pmc_mfma.md > /dev/null
NB=$(python -c "import csv,glob; f=glob.glob('$O/fetch/**/*counter_collection.csv',recursive=True)[0]; print(sum(1 for r in csv.DictReader(open(f)) if 'k_bed_prep_rows' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE'))")
python tools/pmc_traffic.py $O/fetch $O/write $NB $O/${R}_traffic.json 109 1
cat $O/${R}_bench_line.json | cut -c1-400
head -12 $O/${R}_kernel_stats.md
# keep only the summaries (the raw traces are tens of MB)
rm -rf $O/stats $O/seq $O/fetch $O/write $O/mfma
