#!/bin/bash
# Counter passes for the bf16-operand TRAINING kernels at cfg-E (bench.py --config E --precision bf16 --mode train): the SAVE variant of the
# forward chain, the backward chain, the bf16-row scatter-adds and the row-panel kernel; kernel-trace only, one counter group per pass.
# Output: gpurun_out/pmc_bf16_train.txt (copy into profiles/rNN/).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 600 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmct$i -- python $R/bench.py --config E --precision bf16 --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-split-line --no-extras > $R/gpurun_out/pmct$i.log 2>&1
done
cd $R
python - > gpurun_out/pmc_bf16_train.txt <<'PY'
import csv, glob, collections
KEYS = {"edge_chain_bf16_kernel<…, SAVE>": "1, true, 4, false>(mpnhip::EdgeChainBf16Args", "edge_chain_bf16_bwd_kernel": "edge_chain_bf16_bwd_kernel",
        "k_segment_reduce3_b16": "k_segment_reduce3_b16", "wgrad_panel_kernel": "wgrad_panel_kernel", "k_sum_blocks_bf16": "k_sum_blocks_bf16",
        "wgrad_rows16_kernel": "wgrad_rows16_kernel", "gemm_bf16_ring_kernel": "gemm_bf16_ring_kernel"}
print("# tools/pmc_chain_bf16_train.sh: rocprofv3 --kernel-trace --pmc <group> (one group per pass), cfg-E bf16 training step; averages per launch")
for i in range(1, 6):
    fs = glob.glob('gpurun_out/pmct%d/*/*counter_collection.csv' % i)
    if not fs:
        print('no file', i); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        for key, sub in KEYS.items():
            if sub in r['Kernel_Name']:
                acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in sorted(acc.items()):
        for c, v in sorted(d.items()):
            print('%-36s %-28s n=%-4d avg=%.6g' % (k, c, len(v), sum(v) / len(v)))
PY
rm -rf gpurun_out/pmct[1-9]
cat gpurun_out/pmc_bf16_train.txt
