#!/bin/bash
# GPU box: the committed profiles of the default bench command (kernel stats + HBM traffic + SQ counters, separate passes).
# usage: tools/profile.sh <tag>   -> gpurun_out/<tag>_kernel_stats.csv, gpurun_out/<tag>_pmc.json
export PYTHONPATH=/root/repo
REPO=$PWD
TAG=${1:-prof}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_*
BENCH="python3 $REPO/bench.py --no-cpu-baseline $BENCH_ARGS"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_stats -o stats --output-format csv -- $BENCH > $REPO/gpurun_out/${TAG}_stats_run.log 2>&1 || { tail -5 $REPO/gpurun_out/${TAG}_stats_run.log; exit 1; }
cp $(find /tmp/prof_stats -name '*kernel_stats.csv' | head -1) $REPO/gpurun_out/${TAG}_kernel_stats.csv
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES"; do
  d=/tmp/prof_pmc/$(echo $pass | cut -d' ' -f1)
  mkdir -p $d
  export BLCD_LAUNCH_LOG=$d/launch.log    # one line per step_kernel dispatch: pmc_summary.py groups counters by launch length
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace -d $d -o pmc --output-format csv -- $BENCH > $REPO/gpurun_out/${TAG}_pmc_run.log 2>&1 || { tail -5 $REPO/gpurun_out/${TAG}_pmc_run.log; exit 1; }
done
unset BLCD_LAUNCH_LOG
python3 $REPO/tools/pmc_summary.py /tmp/prof_pmc $REPO/gpurun_out/${TAG}_pmc.json
head -5 $REPO/gpurun_out/${TAG}_kernel_stats.csv
