export TMPDIR=/tmp
O=gpurun_out/r04ts; mkdir -p $O; rm -f $O/ab.log
E="python bench.py --config E --precision bf16 --mode fwd --steps 10 --warmup 2 --no-cpu-baseline --no-extras --no-split-line"
ex() { python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'ms %.3f' % d['ms_per_step'], 'chain_us %.1f' % d['roofline']['avg_us'])"; }
$E 2>/dev/null | ex natural_x >> $O/ab.log
$E 2>/dev/null | ex natural_x >> $O/ab.log
cat $O/ab.log
timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -k "bf16_fused_chain_runs or bf16_operand_mode or reads_its_edge" 2>&1 | tail -3
MPNHIP_CHAIN_TS=$O/stamps $E > $O/log.txt 2>&1
python - <<'PY'
import numpy as np
a=np.loadtxt('gpurun_out/r04ts/stamps_bf16fwd.txt',dtype=np.int64)
a=a[(a[:,0]>0)&(a[:,40]>0)]
life=a[:,40]-a[:,0]
print("waves",len(a),"life min/med/max",life.min(),np.median(life),life.max())
def ph(name,i,j): 
    d=a[:,j]-a[:,i]; print("%-28s mean %8.0f med %8.0f min %8d max %8d"%(name,d.mean(),np.median(d),d.min(),d.max()))
ph("prologue (start->sync)",0,1)
ph("H1 tiles (20)",1,21)
ph("e' epilogue",21,22)
ph("cls",22,23)
ph("HF tiles (14)",23,37)
ph("M mask + sync",37,38)
ph("agg rounds",38,39)
ph("tail",39,40)
PY
