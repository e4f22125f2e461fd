export PYTHONPATH=/root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -5
