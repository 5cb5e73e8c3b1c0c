#!/bin/bash
# Counter passes for the bf16-operand chain kernel at cfg-E (bench.py --config E --precision bf16 --mode fwd); kernel-trace only,
# one counter group per pass.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmcb$i -- python $R/bench.py --config E --precision bf16 --mode fwd --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmcb$i.log 2>&1
done
cd $R
python - > gpurun_out/pmc_bf16_fwd.txt <<'PY'
import csv, glob, collections
print('# tools/pmc_chain_bf16.sh: rocprofv3 --kernel-trace --pmc <group> (one group per pass), cfg-E bf16 inference forward; averages per launch')
for i in range(1, 7):
    fs = glob.glob('gpurun_out/pmcb%d/*/*counter_collection.csv' % i)
    if not fs: print('no file', i); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        if 'edge_chain_bf16' not in k and 'k_aggregate' not in k and 'gemm_bf16' not in k: continue
        k = k.replace('void mpnhip::', '').replace('(anonymous namespace)::', '')
        acc[k[:64]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in sorted(acc.items()):
        for c, v in d.items():
            print(k, c, 'n=%d' % len(v), 'avg=%.6g' % (sum(v) / len(v)))
PY
rm -rf gpurun_out/pmcb[1-9]
cat gpurun_out/pmc_bf16_fwd.txt
