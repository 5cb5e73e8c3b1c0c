export TMPDIR=/tmp
O=gpurun_out/r04suite; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu --durations=15 > $O/pytest_gpu.log 2>&1
tail -30 $O/pytest_gpu.log
