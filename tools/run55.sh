export PYTHONPATH=/root/repo
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for W in "Urchin 50000" "Bounce 100000" "Dropbox 100000"; do
  set -- $W
  i=0
  for pass in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH"; do
    i=$((i+1))
    d=/tmp/pmc_ic/$1/$i
    mkdir -p $d
    timeout -k 10 300 rocprofv3 --pmc $pass --kernel-trace -d $d -o pmc --output-format csv -- python3 $REPO/tools/quick_bench.py $1 $2 200 1 > $REPO/gpurun_out/ic_$1_$i.log 2>&1 || { tail -5 $REPO/gpurun_out/ic_$1_$i.log; exit 1; }
  done
done
python3 - <<'PY' > $REPO/gpurun_out/icache_counters.txt
import csv, glob, collections
for w in ('Urchin','Bounce','Dropbox'):
    agg=collections.defaultdict(lambda: [0.0,0])
    for f in glob.glob('/tmp/pmc_ic/%s/*/**/*counter_collection.csv'%w, recursive=True):
        for r in csv.DictReader(open(f)):
            if 'step_kernel' not in r['Kernel_Name']: continue
            a=agg[r['Counter_Name']]; a[0]+=float(r['Counter_Value']); a[1]+=1
    print(w, {k:(v[0], v[1]) for k,v in sorted(agg.items())})
PY
cat $REPO/gpurun_out/icache_counters.txt
