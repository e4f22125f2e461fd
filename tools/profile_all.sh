#!/bin/bash
# GPU box: kernel stats + PMC summaries of every BASELINE workload (committed under profiles/ as r03_*)
set -e
BENCH_ARGS="--steps 2 --warmup 1 --rollouts-per-step 5 --no-configs" tools/profile.sh r03_bounce100k
BENCH_ARGS="--steps 2 --warmup 1 --rollouts-per-step 5 --no-configs --env Dropbox --envs 100000" tools/profile.sh r03_dropbox100k
BENCH_ARGS="--steps 1 --warmup 1 --rollouts-per-step 2 --no-configs --env Urchin --envs 50000" tools/profile.sh r03_urchin50k
BENCH_ARGS="--steps 1 --warmup 1 --rollouts-per-step 2 --no-configs --env LuxoBall --envs 50000" tools/profile.sh r03_luxoball50k
BENCH_ARGS="--steps 1 --warmup 1 --rollouts-per-step 2 --no-configs --env Object2 --envs 200000" tools/profile.sh r03_object2_200k
