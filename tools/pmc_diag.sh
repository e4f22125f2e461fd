#!/bin/bash
# GPU box: where a step_kernel wave's cycles go - issue by instruction class, LDS waits, instruction-cache behaviour.
# usage: BENCH_ARGS="--env Urchin --envs 50000 --steps 1 --warmup 1 --rollouts-per-step 2 --no-configs" tools/pmc_diag.sh <tag>
#   -> gpurun_out/<tag>_diag_pmc.json   (separate --pmc passes, --kernel-trace only)
export PYTHONPATH=/root/repo
REPO=$PWD
TAG=${1:-diag}
mkdir -p $REPO/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/diag_pmc
BENCH="python3 $REPO/bench.py --no-cpu-baseline $BENCH_ARGS"
n=0
for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY" \
            "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM" \
            "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL" \
            "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_FLAT SQ_LDS_BANK_CONFLICT"; do
  n=$((n+1))
  timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace -d /tmp/diag_pmc/p$n -o pmc --output-format csv -- $BENCH > $REPO/gpurun_out/${TAG}_diag_run$n.log 2>&1 || { echo "pass $n failed"; tail -5 $REPO/gpurun_out/${TAG}_diag_run$n.log; }
done
python3 $REPO/tools/pmc_summary.py /tmp/diag_pmc $REPO/gpurun_out/${TAG}_diag_pmc.json
