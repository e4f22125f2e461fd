export PYTHONPATH=/root/repo
timeout -k 10 900 python bench.py > gpurun_out/r04_bench_final.json 2> gpurun_out/r04_bench_final.err || { tail -5 gpurun_out/r04_bench_final.err; exit 1; }
tail -c 600 gpurun_out/r04_bench_final.json
export BLCD_LIB=libboxlcd_hip_wt.so
timeout -k 10 200 python tools/chunk_waves.py Bounce 100000 100 2 || exit 1
timeout -k 10 200 python tools/chunk_waves.py Dropbox 100000 200 1 || exit 1
