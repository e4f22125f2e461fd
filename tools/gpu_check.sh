#!/bin/bash
# GPU box: parity suite, then the default bench line and the other BASELINE configs (short summary on stdout)
export PYTHONPATH=/root/repo
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py "$@" > gpurun_out/bench_x.json 2>gpurun_out/bench_x.err || { tail -5 gpurun_out/bench_x.err; exit 1; }
python -c "import json; d=json.load(open('gpurun_out/bench_x.json')); print('BENCH', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
[ -n "$ALL_ENVS" ] && tools/bench_envs.sh
exit 0
