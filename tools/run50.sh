export PYTHONPATH=/root/repo
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r04_gputest_tw.log 2>&1 || { tail -20 gpurun_out/r04_gputest_tw.log; exit 1; }
tail -3 gpurun_out/r04_gputest_tw.log
bash tools/profile_all.sh
