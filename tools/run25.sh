export PYTHONPATH=/root/repo
SOAK_N=2048 timeout -k 10 1000 python tools/soak.py > gpurun_out/r04_soak_catalogue.log 2>&1; tail -3 gpurun_out/r04_soak_catalogue.log
SOAK_N=65536 SOAK_ENVS=Dropbox,Bounce,Bounce2,Object2,Object3 timeout -k 10 600 python tools/soak.py > gpurun_out/r04_soak_jointfree_65k.log 2>&1; tail -6 gpurun_out/r04_soak_jointfree_65k.log
