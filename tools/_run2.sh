export TMPDIR=/tmp
O=gpurun_out/r04d; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -s -k "bf16" > $O/t1_full.log 2>&1
