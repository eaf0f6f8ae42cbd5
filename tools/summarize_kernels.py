import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
steps=int(sys.argv[2])
tot=0
out=[]
for r in rows:
    t=float(r['TotalDurationNs']); c=int(r['Calls']); tot+=t
    out.append((t/steps/1e3, c/steps, float(r['AverageNs'])/1e3, r['Name'][:70]))
out.sort(reverse=True)
for o in out[:22]: print('%9.1f us/step %7.1f calls/step  avg %8.2f us  %s'%o)
print('total us/step', tot/steps/1e3)
