export PYTHONPATH=/root/repo
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.log 2>&1; rc=$?
tail -3 gpurun_out/gpu_tests.log
[ $rc -ne 0 ] && { grep -n "Error\|assert\|FAILED" gpurun_out/gpu_tests.log | head -20; exit $rc; }
python - <<'PY'
import sys; sys.path.insert(0, '/root/repo')
import bench, json
print(json.dumps({'Bounce-100000': bench.step_loop('Bounce', 100000, 0, 300), 'Dropbox-100000': bench.step_loop('Dropbox', 100000, 0, 300), 'Urchin-50000': bench.step_loop('Urchin', 50000, 0, 40)}, indent=1))
PY
