import csv,sys,glob
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
for r in rows[:int(sys.argv[2]) if len(sys.argv)>2 else 12]:
    print('%-70s %5d avg %8.1f us total %8.2f ms'%(r['Name'][:70], int(r['Calls']), float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
