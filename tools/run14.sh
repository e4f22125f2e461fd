export PYTHONPATH=/root/repo
export BLCD_LIB=libboxlcd_hip_k2.so
QB_NORESET=1 tools/timeline.sh dropbox100k_rest Dropbox 100000 1
grep "step_kernel" gpurun_out/dropbox100k_rest_timeline.txt | tail -10
BLCD_COHORTS=1 tools/timeline.sh dropbox100k_c1 Dropbox 100000 1
grep "step_kernel" gpurun_out/dropbox100k_c1_timeline.txt | tail -6
