export TMPDIR=/tmp
mkdir -p gpurun_out/r04a
python -m pytest tests/test_gpu_loss.py tests/test_gpu_distributed.py -x -q 2>&1 | tail -15 > gpurun_out/r04a/t1.log
python -m pytest tests/test_gpu_modular.py tests/test_gpu_parity.py -x -q -k "batchnorm_trains or counted_barriers or shape_and_index or bf16" 2>&1 | tail -15 > gpurun_out/r04a/t2.log
python -m pytest tests/test_gpu_pinned.py -x -q --durations=8 2>&1 | tail -25 > gpurun_out/r04a/t3.log
rocprofv3 --kernel-trace --stats -d gpurun_out/r04a/prof_cfgE_train -o cfgE -- python bench.py --config E --precision bf16 --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-split-line > gpurun_out/r04a/bench_cfgE_train.log 2>&1
python bench.py --config E --precision bf16 --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-split-line > gpurun_out/r04a/bench_cfgE_train_plain.log 2>&1
ls -R gpurun_out/r04a | head -30
