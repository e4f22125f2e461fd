export PYTHONPATH=/root/repo
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -E "icache|ifetch|inst_cache|SQ_INSTS_|SQ_WAIT|SQ_BUSY|INST_LEVEL|SQC_" | head -80 > $REPO/gpurun_out/avail_counters.txt
wc -l $REPO/gpurun_out/avail_counters.txt
