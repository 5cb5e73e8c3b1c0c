export TMPDIR=/tmp
O=gpurun_out/r04g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "bf16" > $O/t1.log 2>&1
timeout 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_split.py tests/test_gpu_backward.py -q -x > $O/t2.log 2>&1
