bash tools/timeline.sh r04_bounce100k_tw Bounce 100000 3 > /dev/null || exit 1
bash tools/timeline.sh r04_dropbox100k_tw Dropbox 100000 3 > /dev/null || exit 1
cat gpurun_out/r04_bounce100k_tw_timeline.txt gpurun_out/r04_dropbox100k_tw_timeline.txt
