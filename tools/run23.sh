export PYTHONPATH=/root/repo
timeout -k 10 600 python -m pytest tests -x -q -m gpu -k "Dropbox or rest or chunking or cohorts or full_size or every_frame or step_obs" > gpurun_out/gpu_tests_d.log 2>&1; tail -2 gpurun_out/gpu_tests_d.log
timeout -k 10 100 python tools/quick_bench.py Dropbox 100000 200 10 || exit 1
BENCH_ARGS="--steps 2 --warmup 1 --rollouts-per-step 5 --no-configs --env Dropbox --envs 100000" tools/profile.sh r04_dropbox100k > /dev/null 2>&1
tools/timeline.sh dropbox100k_final Dropbox 100000 2 > /dev/null; cat gpurun_out/dropbox100k_final_timeline.txt | head -30
