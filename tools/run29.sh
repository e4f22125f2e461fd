export PYTHONPATH=/root/repo
timeout -k 10 300 python -m pytest tests/test_gpu_api.py -x -q -m gpu -k "rccl or two_ranks" 2>&1 | tail -5
