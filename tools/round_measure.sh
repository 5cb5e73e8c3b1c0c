#!/bin/bash
# Round measurement set (run on the GPU box through gpurun): bench lines, rocprofv3 kernel stats of the default bench
# command, FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, kernel-trace only), tracking-driver numbers.
# Outputs land in gpurun_out/r01/; copy what should be judged into profiles/r01/.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r01
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_r01.json
python $R/bench.py --mode fwd --no-cpu-baseline 2>&1 | tail -1 > $O/bench_fwd_r01.json
python $R/bench.py --config C --mode train --no-cpu-baseline 2>&1 | tail -1 > $O/bench_cfgC_train_r01.json
python $R/bench.py --config C --mode fwd --no-cpu-baseline 2>&1 | tail -1 > $O/bench_cfgC_fwd_r01.json
python $R/bench.py --config D --mode train --no-cpu-baseline 2>&1 | tail -1 > $O/bench_cfgD_train_r01.json
# MPNHIP_PREC_FP32_SPLIT (fp32 results from three-piece bf16 operands in the fused chain kernels): own bench lines + PMC passes
python $R/bench.py --precision fp32_split --no-cpu-baseline 2>&1 | tail -1 > $O/bench_split_r01.json
python $R/bench.py --precision fp32_split --mode fwd --no-cpu-baseline 2>&1 | tail -1 > $O/bench_fwd_split_r01.json
python $R/bench.py --config C --precision fp32_split --mode train --no-cpu-baseline 2>&1 | tail -1 > $O/bench_cfgC_train_split_r01.json
python $R/bench.py --config C --precision fp32_split --mode fwd --no-cpu-baseline 2>&1 | tail -1 > $O/bench_cfgC_fwd_split_r01.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_split -- python $R/bench.py --precision fp32_split --no-cpu-baseline > $O/stats_split.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_split -- python $R/bench.py --precision fp32_split --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/pmc_fetch_split.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_split -- python $R/bench.py --precision fp32_split --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > $O/pmc_write_split.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline --no-split-line > $O/stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-split-line > $O/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-split-line > $O/pmc_write.log 2>&1
cd $R
python tools/track_bench.py 2>/dev/null | tail -1 > $O/track_bench_r01.json
F=$(ls $O/pmc_fetch/*/*counter_collection.csv | head -1); W=$(ls $O/pmc_write/*/*counter_collection.csv | head -1)
python tools/pmc_summary.py $F $W $O/pmc_summary.json
F=$(ls $O/pmc_fetch_split/*/*counter_collection.csv | head -1); W=$(ls $O/pmc_write_split/*/*counter_collection.csv | head -1)
python tools/pmc_summary.py $F $W $O/pmc_summary_split.json
cp $(ls $O/stats_split/*/*kernel_stats.csv | head -1) $O/bench_train_cfgB_split_kernel_stats.csv
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/bench_train_cfgB_kernel_stats.csv
rm -rf $O/stats*/*/*kernel_trace.csv $O/pmc_fetch*/*/*kernel_trace.csv $O/pmc_write*/*/*kernel_trace.csv
head -c 600 $O/bench_r01.json; echo; head -c 300 $O/bench_fwd_r01.json; echo; head -12 $O/bench_train_cfgB_kernel_stats.csv | cut -c1-150
