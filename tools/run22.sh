export PYTHONPATH=/root/repo
for c in 100 200; do echo -n "CHUNK=$c "; BLCD_CHUNK=$c timeout -k 10 100 python tools/quick_bench.py Dropbox 100000 200 10 || exit 1; done
timeout -k 10 100 python tools/quick_bench.py Dropbox 100000 200 10 || exit 1
timeout -k 10 100 python tools/quick_bench.py Dropbox 20000 200 10 || exit 1
BLCD_CHUNK=50 timeout -k 10 100 python tools/quick_bench.py Dropbox 20000 200 10 || exit 1
BENCH_ARGS="--steps 2 --warmup 1 --rollouts-per-step 5 --no-configs --env Dropbox --envs 100000" tools/profile.sh r04_dropbox100k > /dev/null 2>&1
timeout -k 10 600 python bench.py > gpurun_out/bench_r04b.json 2> gpurun_out/bench_r04b.err || { tail -20 gpurun_out/bench_r04b.err; exit 1; }
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_r04b.json'))
print('BENCH', d['value'], 'ms/step', d['ms_per_step'], 'roof', d['roofline']['frac'], 'traffic', d['roofline']['traffic'], d['roofline'].get('traffic_over_algorithmic'))
for k, v in d.get('configs', {}).items():
  print('  ', k, '%.4g' % v['value'], 'sec', round(v['seconds'], 2), 'roof', round(v['roofline']['frac'], 5), 'launch ms', round(v['roofline']['avg_launch_ms'], 2), 'steps/launch', v['roofline']['env_steps_per_env_per_launch'], 'traffic/alg', v['roofline'].get('traffic_over_algorithmic'), (v['roofline'].get('traffic_detail') or {}).get('dropped'))
print('step_loop', {k: v['value'] for k, v in d.get('step_loop', {}).items()})
PY
