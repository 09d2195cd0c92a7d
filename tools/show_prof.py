import csv, sys
rows=list(csv.DictReader(open(f'/root/repo/gpurun_out/prof_{sys.argv[1]}/bench_kernel_stats.csv')))
n=int(sys.argv[2]) if len(sys.argv)>2 else 14
calls=max(int(r['Calls']) for r in rows if 'render_bwd' in r['Name'])
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total kernel ms per step", tot/calls/1e6, " launches/step", sum(int(r['Calls']) for r in rows)/calls)
for r in rows[:n]:
    print(r['Name'][:70].ljust(70), str(int(r['Calls'])/calls)[:5].rjust(6), ("%.1f"%(float(r['AverageNs'])/1e3)).rjust(9), r['Percentage'])
