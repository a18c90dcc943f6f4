#!/bin/bash
# One parameterised job runner for the GPU box (replaces the per-call scripts of rounds 2 and 3).  Every task writes under
# gpurun_out/$OUT (default: the task's name) and prints a short summary; counters are collected in their own runs, never combined
# with API traces.  Usage, several tasks per gpurun call:
#   gpurun -- 'bash tools/gpu_job.sh tests -k full_width; bash tools/gpu_job.sh bench --bt --samples 500000 ...'
#   tests  [pytest args]            python -m pytest tests -q -m gpu <args>
#   bench  [bench.py args]          one bench line -> $O/bench_line.json
#   stats  [bench.py args]          rocprofv3 --kernel-trace --stats of the command -> $O/kernel_stats.md (+ .csv)
#   seq    [bench.py args]          kernel sequence of one step, single pipeline -> $O/batch_sequence.md
#   pmc    "<counters>" [args]      one rocprofv3 --pmc pass (RX= restricts it to kernels matching the regex) -> $O/pmc_<first counter>.md
#   traffic <blocks> <phenos> [args]  FETCH_SIZE and WRITE_SIZE passes -> $O/traffic.json (keyed to the build stamp; copy to profiles/)
#   e2e    <samples> <snps> <phenos>  the C++ driver from files (tools/cli_e2e.py)
#   smoke                           __graft_entry__.smoke()
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
task=$1; shift
O=gpurun_out/${OUT:-$task}
mkdir -p $O
TMO=${TMO:-1500}
case $task in
  tests)
    # paths among the arguments select the tests; without any the whole suite runs
    sel=tests; for a in "$@"; do case $a in tests/*) sel=;; esac; done
    ( time timeout $TMO python -m pytest $sel -q -m gpu "$@" ) > $O/pytest.log 2>&1; tail -15 $O/pytest.log ;;
  bench)
    ( time timeout $TMO python bench.py "$@" ) > $O/bench.log 2>&1
    grep '^{' $O/bench.log | tail -1 > $O/bench_line.json
    python - <<PY
import json
try:
    d = json.load(open("$O/bench_line.json"))
except Exception as e:
    print("no bench line:", e); print(open("$O/bench.log").read()[-3000:]); raise SystemExit
print("ms_per_step", d["ms_per_step"], "value %.4g" % d["value"], "best", d.get("selected_tau_index"), "loco_checksum", d.get("loco_checksum"))
r = d.get("roofline") or {}
print("roofline", r.get("kernel", "")[:40], {k: r.get(k) for k in ("frac", "achieved", "traffic", "avg_launch_ms")})
for k, v in (d.get("kernels") or {}).items():
    print("  %-14s" % k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items()})
print("level1", d.get("level1"))
for k in ("end_to_end_from_files", "cpu_baseline"):
    if d.get(k): print(k, json.dumps(d[k])[:400])
c = d.get("config3_single_gpu")
if c: print("config3", {k: c.get(k) for k in ("ms_per_step", "value", "error")}, (c.get("roofline") or {}).get("frac"), {k: v.get("ms") for k, v in (c.get("kernels") or {}).items()})
s2 = d.get("step2")
if s2:
    print("step2", json.dumps({k: v for k, v in s2.items() if k != "bgen_from_file"})[:600])
    print("step2.bgen_from_file", json.dumps(s2.get("bgen_from_file"))[:1500])
c4 = d.get("config4_level1_binary_traits")
if c4: print("config4_level1", {k: c4.get(k) for k in ("s_per_trait", "converged", "error")}, (c4.get("roofline") or {}).get("frac"), "oracle_check", json.dumps(c4.get("oracle_check"))[:900])
if d.get("bt_oracle_check"): print("bt_oracle_check", json.dumps(d["bt_oracle_check"])[:900])
if d.get("loocv_500k"): print("loocv_500k", json.dumps(d["loocv_500k"])[:1200])
PY
    ;;
  stats)
    timeout $TMO rocprofv3 --kernel-trace --stats --output-format csv -d $O/raw -- python bench.py "$@" > $O/stats.log 2>&1
    python tools/prof_summary.py $O/raw $O/kernel_stats.md > /dev/null
    cp $(find $O/raw -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv 2>/dev/null
    rm -rf $O/raw; head -24 $O/kernel_stats.md ;;
  seq)
    RG_PIPELINES=1 timeout $TMO rocprofv3 --kernel-trace --output-format csv -d $O/raw -- python bench.py "$@" > $O/seq.log 2>&1
    python tools/trace_seq.py $O/raw $O/batch_sequence.md > /dev/null
    rm -rf $O/raw; head -40 $O/batch_sequence.md ;;
  pmc)
    ctr=$1; shift
    name=pmc_$(echo $ctr | cut -d' ' -f1)
    RG_PIPELINES=1 timeout $TMO rocprofv3 --pmc $ctr --kernel-trace ${RX:+--kernel-include-regex "$RX"} --output-format csv -d $O/raw -- python bench.py "$@" > $O/$name.log 2>&1
    python tools/pmc_summary.py $O/raw $O/$name.md > /dev/null
    rm -rf $O/raw; head -30 $O/$name.md ;;
  traffic)
    blocks=$1; phenos=$2; shift 2
    RG_PIPELINES=1 timeout $TMO rocprofv3 --pmc FETCH_SIZE --kernel-trace ${RX:+--kernel-include-regex "$RX"} --output-format csv -d $O/fetch -- python bench.py "$@" > $O/fetch.log 2>&1
    RG_PIPELINES=1 timeout $TMO rocprofv3 --pmc WRITE_SIZE --kernel-trace ${RX:+--kernel-include-regex "$RX"} --output-format csv -d $O/write -- python bench.py "$@" > $O/write.log 2>&1
    python tools/pmc_summary.py $O/fetch $O/pmc_fetch.md > /dev/null
    python tools/pmc_summary.py $O/write $O/pmc_write.md > /dev/null
    NB=$(python -c "import csv,glob; f=glob.glob('$O/fetch/**/*counter_collection.csv',recursive=True)[0]; print(max(1,sum(1 for r in csv.DictReader(open(f)) if 'k_bed_prep_rows' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE')))")
    python tools/pmc_traffic.py $O/fetch $O/write $NB $O/traffic.json $blocks $phenos | cut -c1-400
    rm -rf $O/fetch $O/write ;;
  e2e)
    ( time timeout $TMO python tools/cli_e2e.py "$@" ) > $O/e2e.log 2>&1; tail -30 $O/e2e.log | cut -c1-400 ;;
  smoke)
    python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 ;;
  *) echo "unknown task $task"; exit 2 ;;
esac
