export PYTHONPATH=/root/repo
timeout -k 10 600 python bench.py > gpurun_out/bench_r04a.json 2> gpurun_out/bench_r04a.err || { tail -20 gpurun_out/bench_r04a.err; exit 1; }
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_r04a.json'))
print('BENCH', d['value'], 'ms/step', d['ms_per_step'], 'roof', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], d['roofline'].get('traffic_detail'))
for k, v in d.get('configs', {}).items():
  print('  ', k, '%.4g' % v['value'], 'sec', round(v['seconds'], 2), 'roof', round(v['roofline']['frac'], 5), 'launch ms', round(v['roofline']['avg_launch_ms'], 2), 'steps/launch', v['roofline']['env_steps_per_env_per_launch'])
print('step_loop', json.dumps(d.get('step_loop')))
print('cpu', d.get('cpu_baseline'))
print('parity', d.get('parity'))
PY
