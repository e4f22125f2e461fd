export PYTHONPATH=/root/repo
timeout -k 10 100 python tools/quick_bench.py Object3 100000 200 2 || exit 1
for k in "BLCD_COHORTS=2" "BLCD_COHORTS=3" "BLCD_COHORTS=4" "BLCD_COHORTS=3 BLCD_CHUNK=20" "BLCD_COHORTS=2 BLCD_CHUNK=15" "BLCD_TWO_WIDTHS=24" "BLCD_TWO_WIDTHS=8"; do echo -n "$k :: "; env $k timeout -k 10 100 python tools/quick_bench.py Object2 200000 200 2 || exit 1; done
tools/timeline.sh object2_200k_b Object2 200000 1 > /dev/null
grep "step_kernel" gpurun_out/object2_200k_b_timeline.txt | awk '{print $3, $2}' | awk '{a[$1]=a[$1]" "$2} END{for(k in a) print "queue",k,":",a[k]}' | cut -c1-300
grep "span\|sum" gpurun_out/object2_200k_b_timeline.txt | head -5
