export TMPDIR=/tmp
O=gpurun_out/r04e; mkdir -p $O; rm -f $O/ab.log
B="python bench.py --config E --precision bf16 --mode train --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-split-line"
ex() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); dd=d['details']; print('$1', 'ms %.2f' % d['ms_per_step'], 'fwd_us %.0f' % dd['roofline_fwd_chain']['avg_us'], 'bwd_us %.0f' % dd['roofline_bwd_chain']['avg_us'], 'wg_us %.0f' % dd['roofline_weight_grad']['avg_us'])"; }
$B 2>/dev/null | ex default >> $O/ab.log
MPNHIP_CHAIN_BF16_DEBUG_SKIP=4 $B 2>/dev/null | ex nt_stores >> $O/ab.log
$B 2>/dev/null | ex default2 >> $O/ab.log
MPNHIP_CHAIN_BF16_DEBUG_SKIP=4 $B 2>/dev/null | ex nt_stores2 >> $O/ab.log
cat $O/ab.log
