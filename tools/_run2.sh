export TMPDIR=/tmp
O=gpurun_out/r04f; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_loss.py -q -x > $O/t1.log 2>&1
timeout 600 python bench.py --steps 20 --warmup 10 > $O/bench_default.log 2>$O/bench_default.err
tail -c 2000 $O/bench_default.log
