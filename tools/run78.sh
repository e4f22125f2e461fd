export PYTHONPATH=/root/repo
SOAK_N=16384 timeout -k 10 1000 python tools/soak.py > gpurun_out/r04_soak_catalogue_16k.log 2>&1 || { tail -5 gpurun_out/r04_soak_catalogue_16k.log; exit 1; }
grep -v amdgpu gpurun_out/r04_soak_catalogue_16k.log | tail -20
