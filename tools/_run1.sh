cd /root/repo
python bench.py --no-cpu-baseline > gpurun_out/sp_train.json 2>/dev/null
python bench.py --mode fwd --no-cpu-baseline > gpurun_out/sp_fwd.json 2>/dev/null
