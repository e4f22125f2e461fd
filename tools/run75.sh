export PYTHONPATH=/root/repo
export BENCH_BACKEND=gloo BENCH_DEVICE=0
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/r04_bench_gloo2_rehearsal.json 2> gpurun_out/r04_bench_gloo2.err || { tail -20 gpurun_out/r04_bench_gloo2.err; exit 1; }
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_gloo2_rehearsal.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('metric','value','n_gpus','steps','ms_per_step','scaling')}, list(d.get('configs',{}).keys()), d.get('wire',{}).get('predicted_efficiency_direct') if isinstance(d.get('wire'),dict) else None)
PY
