cd /root/repo
python -m pytest tests/test_gpu_backward.py tests/test_gpu_pinned.py -q -x -k "cfgC or cfgA or generic or encoder or weight_grad" 2>&1 | tail -3 > gpurun_out/nar_t.log
python bench.py --config C --no-cpu-baseline --no-split-line 2>&1 | tail -1 > gpurun_out/nar_C_train.json
python bench.py --config D --no-cpu-baseline --no-split-line 2>&1 | tail -1 > gpurun_out/nar_D_train.json
python bench.py --no-cpu-baseline --no-split-line 2>&1 | tail -1 > gpurun_out/nar_B_train.json
