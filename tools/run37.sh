export PYTHONPATH=/root/repo
timeout -k 10 100 python tools/emit_cost.py Bounce 100000 || exit 1
timeout -k 10 100 python tools/emit_cost.py Dropbox 100000 || exit 1
