#!/bin/bash
# Round measurement set (run on the GPU box through gpurun): bench lines, rocprofv3 kernel stats of the default bench
# command, FETCH_SIZE / WRITE_SIZE PMC passes (separate runs, kernel-trace only), tracking-driver numbers.
# Outputs land in gpurun_out/$RND/; copy what should be judged into profiles/$RND/.
RND=${RND:-r05}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$RND
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log > $O/bench_$RND.json
python $R/bench.py --config E --precision bf16 --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-split-line --no-extras 2>&1 | tail -1 > $O/bench_cfgE_bf16_train_$RND.json
python $R/tools/wgrad_bench.py --check --split > $O/wgrad_bench_split_$RND.txt 2>&1
python $R/tools/gemm_bf16_bench.py > $O/gemm_bf16_bench_$RND.txt 2>&1
python $R/tools/wgrad_rows16_bench.py --check > $O/wgrad_rows16_bench_$RND.txt 2>&1
python $R/tools/gemm_ring_ablate.py > $O/gemm_ring_ablate_$RND.txt 2>&1
python $R/tools/wgrad_bench.py --check > $O/wgrad_bench_fp32_$RND.txt 2>&1
python $R/bench.py --mode fwd --no-cpu-baseline --no-extras 2>&1 | tail -1 > $O/bench_fwd_$RND.json
python $R/bench.py --config C --mode train --no-cpu-baseline 2>&1 | tail -1 > $O/bench_cfgC_train_$RND.json
python $R/bench.py --config C --mode fwd --no-cpu-baseline 2>&1 | tail -1 > $O/bench_cfgC_fwd_$RND.json
python $R/bench.py --config D --mode train --no-cpu-baseline 2>&1 | tail -1 > $O/bench_cfgD_train_$RND.json
python $R/bench.py --config D --mode fwd --steps 300 --warmup 20 --no-cpu-baseline 2>&1 | tail -1 > $O/bench_cfgD_fwd_$RND.json
python $R/bench.py --config E --precision bf16 --mode fwd --steps 10 --warmup 3 2>&1 | tail -1 > $O/bench_cfgE_bf16_fwd_$RND.json
python $R/bench.py --config E --precision fp32 --mode fwd --steps 6 --warmup 2 --no-cpu-baseline --no-split-line 2>&1 | tail -1 > $O/bench_cfgE_fp32_fwd_$RND.json
# the default precision is 'auto' (cfg-B: MPNHIP_PREC_FP32_SPLIT, cfg-C / D: fp32 MFMAs); fp32 MFMAs everywhere: own bench lines
python $R/bench.py --precision fp32 --no-cpu-baseline --no-split-line --no-extras 2>&1 | tail -1 > $O/bench_fp32mfma_$RND.json
python $R/bench.py --precision fp32 --mode fwd --no-cpu-baseline --no-split-line --no-extras 2>&1 | tail -1 > $O/bench_fwd_fp32mfma_$RND.json
# kernel stats of the default command and of the cfg-E line
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python $R/bench.py --no-cpu-baseline --no-split-line --no-extras --no-forward-rate > $O/stats.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_fwd -- python $R/bench.py --mode fwd --no-cpu-baseline --no-split-line > $O/stats_fwd.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfgE -- python $R/bench.py --config E --precision bf16 --mode fwd --steps 10 --warmup 3 --no-cpu-baseline > $O/stats_cfgE.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfgC -- python $R/bench.py --config C --no-cpu-baseline --no-split-line > $O/stats_cfgC.log 2>&1
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfgE_train -- python $R/bench.py --config E --precision bf16 --mode train --steps 10 --warmup 4 --no-cpu-baseline --no-split-line --no-extras --no-forward-rate > $O/stats_cfgE_train.log 2>&1
timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_cfgD -- python $R/bench.py --config D --no-cpu-baseline --no-split-line --no-extras > $O/stats_cfgD.log 2>&1
# HBM traffic (PMC): FETCH_SIZE and WRITE_SIZE in separate passes
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_Et -- python $R/bench.py --config E --precision bf16 --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-split-line --no-extras > $O/pmc_fetch_Et.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_Et -- python $R/bench.py --config E --precision bf16 --mode train --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-split-line --no-extras > $O/pmc_write_Et.log 2>&1
# (--no-forward-rate: the passes hold TRAINING launches only -- with the forward-rate leg the chain kernel's average mixed 276 inference
# launches into 84 training ones, round 6)
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-split-line --no-extras --no-forward-rate > $O/pmc_fetch.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-split-line --no-extras --no-forward-rate > $O/pmc_write.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch32 -- python $R/bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-split-line --no-extras --no-forward-rate > $O/pmc_fetch32.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write32 -- python $R/bench.py --precision fp32 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-split-line --no-extras --no-forward-rate > $O/pmc_write32.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_E -- python $R/bench.py --config E --precision bf16 --mode fwd --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_fetch_E.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_E -- python $R/bench.py --config E --precision bf16 --mode fwd --steps 3 --warmup 1 --no-cpu-baseline --no-roofline > $O/pmc_write_E.log 2>&1
cd $R
python tools/track_bench.py 2>/dev/null | tail -1 > $O/track_bench_$RND.json
python tools/step_timeline.py $O/stats > $O/step_timeline_cfgB_train.txt 2>/dev/null
python tools/step_timeline.py $O/stats_cfgE_train > $O/step_timeline_cfgE_train.txt 2>/dev/null
F=$(ls $O/pmc_fetch/*/*counter_collection.csv | head -1); W=$(ls $O/pmc_write/*/*counter_collection.csv | head -1)
python tools/pmc_summary.py $F $W $O/pmc_summary_split.json > /dev/null
F=$(ls $O/pmc_fetch32/*/*counter_collection.csv | head -1); W=$(ls $O/pmc_write32/*/*counter_collection.csv | head -1)
python tools/pmc_summary.py $F $W $O/pmc_summary.json > /dev/null
F=$(ls $O/pmc_fetch_E/*/*counter_collection.csv | head -1); W=$(ls $O/pmc_write_E/*/*counter_collection.csv | head -1)
python tools/pmc_summary.py $F $W $O/pmc_summary_cfgE.json > /dev/null
F=$(ls $O/pmc_fetch_Et/*/*counter_collection.csv | head -1); W=$(ls $O/pmc_write_Et/*/*counter_collection.csv | head -1)
python tools/pmc_summary.py $F $W $O/pmc_summary_cfgE_train.json > /dev/null
cp $(ls $O/stats_cfgE_train/*/*kernel_stats.csv | head -1) $O/bench_train_cfgE_bf16_kernel_stats.csv
cp $(ls $O/stats_cfgD/*/*kernel_stats.csv | head -1) $O/bench_train_cfgD_kernel_stats.csv
[ -x build/micro/store_pattern ] && { ./build/micro/store_pattern 400000 20; ./build/micro/store_pattern 50000 50; } > $O/store_pattern.txt 2>&1
cp $(ls $O/stats/*/*kernel_stats.csv | head -1) $O/bench_train_cfgB_kernel_stats.csv
cp $(ls $O/stats_fwd/*/*kernel_stats.csv | head -1) $O/bench_fwd_cfgB_kernel_stats.csv
cp $(ls $O/stats_cfgE/*/*kernel_stats.csv | head -1) $O/bench_fwd_cfgE_bf16_kernel_stats.csv
cp $(ls $O/stats_cfgC/*/*kernel_stats.csv | head -1) $O/bench_train_cfgC_kernel_stats.csv
rm -rf $O/stats*/ $O/pmc_fetch*/ $O/pmc_write*/
head -c 500 $O/bench_$RND.json; echo; head -c 300 $O/bench_fwd_$RND.json; echo; head -8 $O/bench_train_cfgB_kernel_stats.csv | cut -c1-150
