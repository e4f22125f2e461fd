export PYTHONPATH=/root/repo
export BLCD_LIB=libboxlcd_hip_k2.so
echo "== swapped streams"
BLCD_COHORT_SWAP=1 QB_NORESET=1 tools/timeline.sh dropbox100k_rest_swap Dropbox 100000 1 > /dev/null
grep "step_kernel" gpurun_out/dropbox100k_rest_swap_timeline.txt | tail -9
