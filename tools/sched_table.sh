#!/bin/bash
# One table of the environment-level scheduling measurements of DESIGN.md 4.4 (plain rollout vs chunk passes vs asynchronous
# rollouts vs in-wave batching, each at the setting that was best in the sweeps of tools/yield_bench.py / async_bench.sh /
# wb_bench.sh).  usage: bash tools/sched_table.sh > profiles/r03_sched_measurements.log
export PYTHONPATH=${GRAFT_REPO_ROOT:-/root/repo}
one() { timeout -k 10 300 python tools/yield_bench.py --one $1 $2 200 2>&1 | grep -v amdgpu | tail -1; }
run() {  # label, env assignments..., then workload
  local label=$1; shift
  local envs=(); while [[ $1 == *=* ]]; do envs+=("$1"); shift; done
  echo -n "$label :: "; env "${envs[@]}" bash -c "$(declare -f one); one $1 $2"
}
for spec in "Dropbox 100000 2 2" "Object2 200000 2 2" "Urchin 50000 3 4" "LuxoBall 50000 3 4"; do
  set -- $spec
  run "plain            " X=0 $1 $2
  if [ $1 = Dropbox ] || [ $1 = Object2 ]; then
    run "chunk passes p=2 lanes<=32 chunk=20" BLCD_YIELD_PASSES=2 BLCD_YIELD_LANES=32 BLCD_CHUNK=20 $1 $2
  fi
  run "asynchronous k=$3 lanes<=8 " BLCD_ASYNC=$3 BLCD_YIELD_LANES=8 $1 $2
  run "in-wave batching r=$4 lanes<=8" BLCD_WAVE_BATCH=$4 BLCD_YIELD_LANES=8 $1 $2
done
