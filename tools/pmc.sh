#!/bin/bash
# GPU box: SQ issue/wait counters of one bench rollout (own pass, no traces besides --kernel-trace)
export PYTHONPATH=/root/repo
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_sq
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU \
  --kernel-trace -d /tmp/pmc_sq -o pmc --output-format csv -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline $BENCH_ARGS > $REPO/gpurun_out/pmc_run.log 2>&1 || { tail -5 $REPO/gpurun_out/pmc_run.log; exit 1; }
python3 $REPO/tools/pmc_summary.py /tmp/pmc_sq $REPO/gpurun_out/pmc_sq.json
