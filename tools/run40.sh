export PYTHONPATH=/root/repo
# SQ counters of the jointed class at one and at two waves per SIMD (the negative result of DESIGN 4.7), 100 000 environments
for lib in libboxlcd_hip.so libboxlcd_hip_w2.so; do
  export BLCD_LIB=$lib
  tag=r04_urchin100k_$(echo $lib | sed 's/libboxlcd_hip//; s/\.so//; s/_//')x
  BENCH_ARGS="--steps 1 --warmup 1 --rollouts-per-step 1 --no-configs --env Urchin --envs 100000" tools/profile.sh $tag > /dev/null 2>&1
  python3 - $tag <<'PY'
import json, sys
d = json.load(open(f'gpurun_out/{sys.argv[1]}_pmc.json'))
k = [x for x in d if 'step_kernel' in x][0]
c = d[k]
w = c['SQ_WAVE_CYCLES']['mean']
print(sys.argv[1], 'waves', c['SQ_WAVES']['mean'], 'VALU busy %.3f waiting %.3f lanes/VALU %.1f FETCH GB %.1f WRITE GB %.1f' % (c['SQ_ACTIVE_INST_VALU']['mean'] / w, c['SQ_WAIT_ANY']['mean'] / w, c['SQ_THREAD_CYCLES_VALU']['mean'] / c['SQ_ACTIVE_INST_VALU']['mean'], c['FETCH_SIZE']['mean'] * 1024 / 1e9, c['WRITE_SIZE']['mean'] * 1024 / 1e9))
PY
  grep step_kernel gpurun_out/${tag}_kernel_stats.csv | cut -d, -f2-4 | cut -c1-80
done
