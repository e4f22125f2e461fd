export PYTHONPATH=/root/repo
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small_batches or two_wave_widths or cohorts" 2>&1 | tail -6
