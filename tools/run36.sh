export PYTHONPATH=/root/repo
timeout -k 10 600 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err || { tail -20 gpurun_out/r04_bench_final.err; exit 1; }
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_bench_final.json'))
print('BENCH', d['value'], 'roof', d['roofline']['frac'], 'traffic/alg', d['roofline'].get('traffic_over_algorithmic'))
for k, v in d.get('configs', {}).items():
  print('  ', k, '%.4g' % v['value'], 'roof', round(v['roofline']['frac'], 5), 'traffic/alg', v['roofline'].get('traffic_over_algorithmic'), (v['roofline'].get('traffic_detail') or {}).get('dropped'))
PY
