export PYTHONPATH=/root/repo
export BENCH_BACKEND=gloo BENCH_DEVICE=0 MASTER_PORT=29577
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29577 bench.py --gpus 2 --steps 1 --warmup 0 --rollouts-per-step 2 --envs 50000 > gpurun_out/bench_gloo2.json 2> gpurun_out/bench_gloo2.err || { tail -30 gpurun_out/bench_gloo2.err; exit 1; }
python - <<'PY'
import json
d = json.load(open('gpurun_out/bench_gloo2.json'))
print(json.dumps({k: d[k] for k in ('value', 'n_gpus', 'stepping_only', 'wire')}, indent=1))
print(json.dumps(d.get('configs'), indent=1))
PY
