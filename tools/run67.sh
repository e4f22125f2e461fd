export PYTHONPATH=/root/repo
timeout -k 10 600 python -m pytest tests/test_gpu_api.py -m gpu -x -q 2>&1 | tail -6
timeout -k 10 600 python bench.py --steps 5 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], {k:(v['value'], v['ms_per_call'], v['async']['value'], v['async']['ms_per_call'], v['async']['faults']) for k,v in d.get('step_loop',{}).items()})"
