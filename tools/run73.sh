export PYTHONPATH=/root/repo
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputest_final.log 2>&1 || { tail -20 gpurun_out/r04_gputest_final.log; exit 1; }
tail -2 gpurun_out/r04_gputest_final.log
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err || { tail -5 gpurun_out/r04_bench_final.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_final.json').read().strip().splitlines()[-1])
print(d['value'], {k:v['value'] for k,v in d['configs'].items()}, {k:(v['value'], v['async']['value'], v['inline']['value']) for k,v in d['step_loop'].items()})
PY
python __graft_entry__.py smoke 2>&1 | tail -1
