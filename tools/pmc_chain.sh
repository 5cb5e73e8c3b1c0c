#!/bin/bash
# (the TA_FLAT_* / TA_*_STALLED_* group aborted rocprofv3 on this pool and hung the call: do not add it back)
# PMC passes for the fused chain kernels (each counter group in its own run; kernel-trace only, as gpurun requires)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
MODE=${1:-fwd}
i=0
for grp in ${PMC_GROUPS:+"$PMC_GROUPS"} "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" "TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TCP_TA_TCP_STATE_READ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmcc$i -- python $R/bench.py --config B --mode $MODE --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $R/gpurun_out/pmcc$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for i in range(1, 7):
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
