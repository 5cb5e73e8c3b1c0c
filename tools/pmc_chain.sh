#!/bin/bash
# PMC passes for the fused chain kernels (each counter group in its own run; kernel-trace only, as gpurun requires)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MODE=${1:-fwd}
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmcc$i -- python $R/bench.py --config B --mode $MODE --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmcc$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for i in range(1, 6):
    fs = glob.glob('gpurun_out/pmcc%d/*/*counter_collection.csv' % i)
    if not fs: print('no file', i); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        if 'edge_chain' not in k: continue
        acc[k[:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        for c, v in d.items():
            print(k, c, 'n=%d' % len(v), 'avg=%.4g' % (sum(v) / len(v)))
PY
