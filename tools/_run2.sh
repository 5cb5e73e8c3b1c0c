export TMPDIR=/tmp
O=gpurun_out/r04b; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -q -k "bf16" 2>&1 | tail -60 > $O/t1.log
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 tests/dist_train_check.py 6 32 nccl1 > $O/nccl1.log 2>&1
timeout 600 python -m pytest tests/test_gpu_distributed.py -q -k "invalid or rccl" 2>&1 | tail -15 > $O/t2.log
