export PYTHONPATH=/root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_api.py tests/test_gpu_goal.py -m gpu -x -q 2>&1 | tail -4
python tools/step_loop_probe.py Bounce 100000 300
python tools/step_loop_probe.py Urchin 50000 40
python tools/step_loop_probe.py Dropbox 100000 300
