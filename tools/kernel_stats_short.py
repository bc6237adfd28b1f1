import csv,glob,sys,re
f=glob.glob(sys.argv[1]+'/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    n=r['Name']
    m=re.search(r'INS0_\d+(\w+?Args)', n)
    short = m.group(1) if m else n.split('(')[0][-60:]
    if float(r['Percentage'])>1.0:
        print(f"{short:45s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms  {r['Percentage']}%")
