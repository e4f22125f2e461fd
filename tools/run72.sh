export PYTHONPATH=/root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -5
(python tools/step_loop_probe.py Bounce 100000 300; python tools/step_loop_probe.py Dropbox 100000 300; python tools/step_loop_probe.py Urchin 50000 40) > gpurun_out/r04_step_loop_probe.txt
cat gpurun_out/r04_step_loop_probe.txt
