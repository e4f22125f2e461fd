export PYTHONPATH=/root/repo
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputest_final.log 2>&1 || { tail -20 gpurun_out/r04_gputest_final.log; exit 1; }
tail -2 gpurun_out/r04_gputest_final.log
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err || { tail -5 gpurun_out/r04_bench_final.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_final.json').read().strip().splitlines()[-1])
print(d['value'], {k:v['value'] for k,v in d['configs'].items()}, {k:(v['value'], v['async']['value']) for k,v in d['step_loop'].items()})
PY
(python tools/step_loop_probe.py Bounce 100000 300; python tools/step_loop_probe.py Dropbox 100000 300; python tools/step_loop_probe.py Urchin 50000 40) > gpurun_out/r04_step_loop_probe.txt
cat gpurun_out/r04_step_loop_probe.txt
python __graft_entry__.py smoke 2>&1 | tail -1
