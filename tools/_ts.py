import numpy as np, sys
a = np.loadtxt(sys.argv[1], dtype=np.int64)
a = a[a[:, 0] > 0]
print('waves', len(a))
t0 = a[:, 0].min()
names = ['start->chunk0', 'phase1 loop', 'Pc add/save', 'phase2', 'phase3', 'phase4', 'phase5 mfma', 'epilogue']
idx = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 6), (6, 7), (7, 8)]
full = a[a[:, 8] > 0]
print('flow waves', len(full))
tot = full[:, 8] - full[:, 0]
print('wave life cycles: min %d med %d max %d' % (tot.min(), np.median(tot), tot.max()))
print('kernel span cycles', a[:, [0,1,2,3,4,5]].max() - t0, 'end', full[:, 8].max() - t0, 'start spread', a[:, 0].max() - t0)
hw = full[:, 12]
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3; 
for n, (i, j) in zip(names, idx):
    d = full[:, j] - full[:, i]
    print('%-14s mean %8.0f  min %8d  max %8d' % (n, d.mean(), d.min(), d.max()))
d = full[:, 10] - full[:, 9]
print('commit (i=3)   mean %8.0f min %d max %d' % (d.mean(), d.min(), d.max()))
# split by lifetime quantiles
q = np.argsort(tot)
for nm, sel in (('fastest 25%', q[:len(q)//4]), ('slowest 25%', q[-len(q)//4:])):
    print(nm, 'life', tot[sel].mean(), [int((full[sel, j] - full[sel, i]).mean()) for i, j in idx])
