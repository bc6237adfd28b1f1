import csv,glob,sys
f=glob.glob(sys.argv[1]+'/*/*kernel_trace.csv')[0]
rows=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in csv.DictReader(open(f))]
rows.sort()
# split into calls by gaps > 2 ms
calls=[];cur=[rows[0]]
for r in rows[1:]:
    if r[0]-max(x[1] for x in cur) > 1_500_000: calls.append(cur);cur=[r]
    else: cur.append(r)
calls.append(cur)
for c in calls:
    if len(c)<30: continue
    t0=c[0][0];t1=max(x[1] for x in c)
    # union
    busy=0;e=t0
    for s_,e_,_ in c:
        if e_<=e: continue
        busy+=e_-max(s_,e);e=e_
    names={}
    for s_,e_,n in c:
        k='Lo' if 'UsacLoArgs' in n else 'Check' if 'UsacCheck' in n else 'Roots' if 'RootsArgs' in n else 'Solve' if 'SolveArgs' in n else 'other'
        names[k]=names.get(k,0)+(e_-s_)
    print(f"call: span {(t1-t0)/1e6:.2f} ms, GPU busy (union) {busy/1e6:.2f} ms = {busy/(t1-t0):.2f}; kernels {len(c)}; sums ms {{k: round(v/1e6,2) for k,v in names.items()}}".replace("{{","{").replace("}}","}"), {k: round(v/1e6,2) for k,v in names.items()})
