export PYTHONPATH=/root/repo
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err || { tail -5 gpurun_out/r04_bench_final.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_final.json').read().strip().splitlines()[-1])
print(d['value'], d['roofline']['traffic'] is not None, {k:v['value'] for k,v in d['configs'].items()}, {k:(v['value'], v['async']['value'], v['inline']['value']) for k,v in d['step_loop'].items()})
PY
