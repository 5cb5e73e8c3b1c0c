#!/bin/bash
# Timing ablation of node_chain_kernel (build: touch mpntrackseg_amd/csrc/node_chain.hip && make EXTRA=-DMPNHIP_NODE_FWD_DEBUG).
# MPNHIP_NODE_FWD_DEBUG bits: 1 no message-row loads, 2 return after the aggregation, 4 return after the node update, 8 no MFMAs in the
# projections, 16 no weight-unit loads there, 32 no P' stores.  Results are wrong with any bit set: timing only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ncab; mkdir -p $O
for dbg in ${NC_BITS:-0 1 2 3 4 8 16 24 32 56}; do
  MPNHIP_NODE_FWD_DEBUG=$dbg timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$dbg -- python $R/bench.py --mode fwd --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-split-line --no-extras > $O/s$dbg.log 2>&1
  f=$(ls $O/s$dbg/*/*kernel_stats.csv | head -1)
  python3 - $f $dbg <<'PY' | tee -a $O/node_chain_ablation.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'node_chain_kernel' in r['Name'] or 'edge_chain_kernel' in r['Name']:
        print('debug=%s %-28s calls %s avg_us %.2f min_us %.2f max_us %.2f' % (sys.argv[2], r['Name'].split('(')[0][-28:], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
  rm -rf $O/s$dbg
done
