export PYTHONPATH=/root/repo
SOAK_N=2048 timeout -k 10 500 python tools/soak.py > gpurun_out/r04_soak_catalogue.log 2>&1 || { tail -5 gpurun_out/r04_soak_catalogue.log; exit 1; }
tail -1 gpurun_out/r04_soak_catalogue.log
SOAK_N=2048 timeout -k 10 500 python tools/soak_frames.py > gpurun_out/r04_soak_every_frame.log 2>&1 || { tail -5 gpurun_out/r04_soak_every_frame.log; exit 1; }
tail -1 gpurun_out/r04_soak_every_frame.log
SOAK_N=100000 SOAK_ENVS=Dropbox,Bounce,Bounce2,Object2,Object3 timeout -k 10 600 python tools/soak.py > gpurun_out/r04_soak_jointfree_100k.log 2>&1 || { tail -5 gpurun_out/r04_soak_jointfree_100k.log; exit 1; }
tail -6 gpurun_out/r04_soak_jointfree_100k.log
