#!/bin/bash
# Counter passes for the kernels the HEADLINE runs (bench.py default: cfg-B, MPNHIP_PREC_FP32_SPLIT, training step): forward chain,
# backward chain, row-panel weight gradients, node chain (fwd / bwd), the backward's scatter-add.  kernel-trace only, one counter
# group per pass, the program directly after `--`.  Output: gpurun_out/pmc_cfgB_split.txt (+ .json) -> copy into profiles/rNN/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmcb$i -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-split-line --no-extras --no-forward-rate ${PMC_BENCH_ARGS} > $R/gpurun_out/pmcb$i.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections, json
KEYS = {"edge_chain_kernel": "edge_chain_kernel<", "edge_chain_bwd_kernel": "edge_chain_bwd_kernel<", "wgrad_panel_kernel": "wgrad_panel_kernel",
        "wgrad_panel_narrow_kernel": "wgrad_panel_narrow_kernel", "node_chain_kernel": "node_chain_kernel<", "node_chain_bwd_kernel": "node_chain_bwd_kernel<",
        "k_segment_reduce3": "k_segment_reduce3", "k_sum_blocks": "k_sum_blocks", "wgrad_reduce_kernel": "wgrad_reduce_kernel"}
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for i in range(1, 5):
    fs = glob.glob('gpurun_out/pmcb%d/*/*counter_collection.csv' % i)
    if not fs:
        print('no file', i); continue
    for r in csv.DictReader(open(fs[0])):
        for key, sub in KEYS.items():
            if sub in r['Kernel_Name']:
                acc[key][r['Counter_Name']].append(float(r['Counter_Value']))
                break
lines = ["# tools/pmc_cfgB.sh: rocprofv3 --kernel-trace --pmc <group> (one group per pass) over `python bench.py` (cfg-B, fp32_split, training step); averages per launch"]
summ = {}
for k, d in sorted(acc.items()):
    avg = {c: sum(v) / len(v) for c, v in d.items()}
    for c, v in sorted(d.items()):
        lines.append('%-28s %-28s n=%-4d avg=%.6g' % (k, c, len(v), avg[c]))
    s = {}
    if avg.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in avg:
        # SQ_VALU_MFMA_BUSY_CYCLES: cycles summed over the SIMDs; GRBM_GUI_ACTIVE: summed over the 8 XCDs (MI355X guide) -> fraction of the
        # 1,024 SIMD-cycles of the launch during which an MFMA was executing
        s["mfma_busy_frac"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * avg["GRBM_GUI_ACTIVE"] / 8.0)
    if avg.get("SQ_WAVE_CYCLES"):
        s["wait_inst_any_over_wave_cycles"] = avg.get("SQ_WAIT_INST_ANY", 0.0) / avg["SQ_WAVE_CYCLES"]
        if avg.get("SQ_BUSY_CYCLES"):
            s["sq_busy_cycles"] = avg["SQ_BUSY_CYCLES"]
    if avg.get("SQ_ACTIVE_INST_LDS"):
        s["lds_bank_conflict_over_active_lds"] = avg.get("SQ_LDS_BANK_CONFLICT", 0.0) / avg["SQ_ACTIVE_INST_LDS"]
    if avg.get("TCC_REQ_sum"):
        s["l2_hit_rate"] = avg.get("TCC_HIT_sum", 0.0) / max(avg.get("TCC_HIT_sum", 0.0) + avg.get("TCC_MISS_sum", 0.0), 1.0)
    summ[k] = s
    lines.append('%-28s %s' % (k, '  '.join('%s=%.4f' % kv for kv in s.items() if kv[0] != "sq_busy_cycles")))
open('gpurun_out/pmc_cfgB_split.txt', 'w').write('\n'.join(lines) + '\n')
json.dump(summ, open('gpurun_out/pmc_cfgB_split.json', 'w'), indent=1)
print('\n'.join(l for l in lines if '=' in l and 'n=' not in l))
PY
rm -rf gpurun_out/pmcb[1-9]
