#!/bin/bash
# SQ / TCC counter passes for the weight-gradient product kernel in isolation (tools/wgrad_bench.py); kernel-trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ARGS=${WG_ARGS:---iters 3}     # e.g. WG_ARGS="--iters 3 --split --only 1,2,4"
KERN=${WG_KERNEL:-gemm_tn_kernel}   # wgrad_panel_kernel for --split
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "GRBM_GUI_ACTIVE FETCH_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmcw$i -- python $R/tools/wgrad_bench.py $ARGS > $R/gpurun_out/pmcw$i.log 2>&1
done
cd $R
KERN=$KERN python - <<'PY'
import csv, glob, collections, os
for i in range(1, 6):
    fs = glob.glob('gpurun_out/pmcw%d/*/*counter_collection.csv' % i)
    if not fs: print('no file', i); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(fs[0])):
        k = r['Kernel_Name']
        if os.environ['KERN'] not in k: continue
        acc[k[:40] + ' grid ' + r.get('Grid_Size', '?')][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in sorted(acc.items()):
        for c, v in d.items():
            print(k, c, 'n=%d' % len(v), 'avg=%.4g' % (sum(v) / len(v)))
PY
