export TMPDIR=/tmp
O=gpurun_out/r04j; mkdir -p $O; rm -f $O/rep.log
for i in 1 2; do
timeout 600 python -m pytest tests/test_gpu_parity.py -q -k "bf16" 2>&1 | grep -E "passed|failed|^FAILED" >> $O/rep.log
done
cat $O/rep.log
