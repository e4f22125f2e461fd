#!/bin/bash
export PYTHONPATH=/root/repo
REPO=$PWD
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lat_pmc
BENCH="python3 $REPO/bench.py --no-cpu-baseline --env Urchin --envs 50000 --steps 1 --warmup 1 --rollouts-per-step 2 --no-configs"
timeout -k 10 300 rocprofv3 --pmc SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_FLAT SQ_INSTS_LDS SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d /tmp/lat_pmc/p1 -o pmc --output-format csv -- $BENCH > $REPO/gpurun_out/lat_run.log 2>&1 || tail -5 $REPO/gpurun_out/lat_run.log
python3 $REPO/tools/pmc_summary.py /tmp/lat_pmc $REPO/gpurun_out/urchin50k_lat_pmc.json
