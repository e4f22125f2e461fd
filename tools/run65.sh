export PYTHONPATH=/root/repo
timeout -k 10 200 python tools/torch_order_check.py on || exit 1
timeout -k 10 200 python tools/torch_order_check.py off || exit 1
timeout -k 10 600 python bench.py --steps 5 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], {k:(v['value'], v['ms_per_call']) for k,v in d.get('step_loop',{}).items()})"
