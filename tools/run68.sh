export PYTHONPATH=/root/repo
REPO=$PWD
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputest_final.log 2>&1 || { tail -20 gpurun_out/r04_gputest_final.log; exit 1; }
tail -2 gpurun_out/r04_gputest_final.log
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err || { tail -5 gpurun_out/r04_bench_final.err; exit 1; }
python tools/step_loop_probe.py Bounce 100000 300 > gpurun_out/r04_step_loop_probe.txt
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/sl -o sl --output-format csv -- python3 $REPO/tools/step_loop_probe.py Bounce 100000 300 > /dev/null 2>&1 || exit 1
f=$(find /tmp/sl -name '*kernel_stats.csv' | head -1)
python3 - "$f" >> $REPO/gpurun_out/r04_step_loop_probe.txt <<'PY'
import csv, sys
print('rocprofv3 --kernel-trace --stats of the same script (both loops, 620 steps):')
for r in list(csv.DictReader(open(sys.argv[1])))[:8]:
    print('  %-60s calls %5s avg %8.1f us  %5.1f %%' % (r['Name'].split('(')[0][:60], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
cat $REPO/gpurun_out/r04_step_loop_probe.txt
