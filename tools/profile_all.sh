#!/bin/bash
# GPU box: kernel stats + PMC summaries of every BASELINE workload and the largest scene class (committed under profiles/ as r04_*)
set -e
R=${R:-r04}
BENCH_ARGS="--steps 2 --warmup 1 --rollouts-per-step 5 --no-configs" tools/profile.sh ${R}_bounce100k
BENCH_ARGS="--steps 2 --warmup 1 --rollouts-per-step 5 --no-configs --env Dropbox --envs 100000" tools/profile.sh ${R}_dropbox100k
BENCH_ARGS="--steps 1 --warmup 1 --rollouts-per-step 2 --no-configs --env Urchin --envs 50000" tools/profile.sh ${R}_urchin50k
BENCH_ARGS="--steps 1 --warmup 1 --rollouts-per-step 2 --no-configs --env LuxoBall --envs 50000" tools/profile.sh ${R}_luxoball50k
BENCH_ARGS="--steps 1 --warmup 1 --rollouts-per-step 2 --no-configs --env Object2 --envs 200000" tools/profile.sh ${R}_object2_200k
BENCH_ARGS="--steps 1 --warmup 0 --rollouts-per-step 1 --no-configs --env Crab --envs 20000" tools/profile.sh ${R}_crab20k
