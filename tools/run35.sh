export PYTHONPATH=/root/repo
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r04_gputest_final.log 2>&1; rc=$?
tail -3 gpurun_out/r04_gputest_final.log
[ $rc -ne 0 ] && { grep -n "Error\|assert \|FAILED" gpurun_out/r04_gputest_final.log | head -20; exit $rc; }
python -c "import __graft_entry__ as g; g.smoke()" || exit 1
bash tools/profile_all.sh > gpurun_out/profile_all.log 2>&1; tail -2 gpurun_out/profile_all.log
timeout -k 10 600 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err || { tail -20 gpurun_out/r04_bench_final.err; exit 1; }
python - <<'PY'
import json
d = json.load(open('gpurun_out/r04_bench_final.json'))
print('BENCH', d['value'], 'ms/step', d['ms_per_step'], 'roof', d['roofline']['frac'], 'traffic/alg', d['roofline'].get('traffic_over_algorithmic'))
for k, v in d.get('configs', {}).items():
  print('  ', k, '%.4g' % v['value'], 'roof', round(v['roofline']['frac'], 5), 'launch ms', round(v['roofline']['avg_launch_ms'], 2), 'steps/launch', v['roofline']['env_steps_per_env_per_launch'], 'traffic/alg', v['roofline'].get('traffic_over_algorithmic'), (v['roofline'].get('traffic_detail') or {}).get('dropped'))
print('step_loop', {k: v['value'] for k, v in d.get('step_loop', {}).items()})
PY
