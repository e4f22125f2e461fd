export PYTHONPATH=/root/repo
timeout -k 10 600 python bench.py --steps 2 --warmup 1 > gpurun_out/bench_check.json 2> gpurun_out/bench_check.err || { tail -20 gpurun_out/bench_check.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/bench_check.json')); print(d['value'], list(d['configs']), {k:('error' in v) for k,v in d['configs'].items()}, d['step_loop'].keys(), d['roofline']['traffic'] is not None)"
export BENCH_BACKEND=gloo BENCH_DEVICE=0 MASTER_PORT=29578
timeout -k 10 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29578 bench.py --gpus 2 --steps 1 --warmup 0 --rollouts-per-step 2 --envs 50000 > gpurun_out/bench_gloo2.json 2> gpurun_out/bench_gloo2.err || { tail -30 gpurun_out/bench_gloo2.err; exit 1; }
python -c "
import json; d=json.load(open('gpurun_out/bench_gloo2.json')); print(d['value'], d['n_gpus'], list(d['configs']), {k:('error' in v) for k,v in d['configs'].items()}, 'wire' in d)"
