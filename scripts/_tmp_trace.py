import csv,glob,sys,collections
d=sys.argv[1]
k=glob.glob(d+'/*kernel_trace.csv')[0]
ev=[]
for r in csv.DictReader(open(k)):
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name'].split('(')[0][-58:],r['Queue_Id'],r['Grid_Size_X'],r['Workgroup_Size_X']))
ev.sort()
ev=[e for e in ev if e[0] > ev[-1][0]-int(float(sys.argv[2])*1e9)]
q=collections.Counter(e[3] for e in ev)
print('queues',q)
qsel=sys.argv[3] if len(sys.argv)>3 else q.most_common(1)[0][0]
sel=[e for e in ev if e[3]==qsel][-int(sys.argv[4]) if len(sys.argv)>4 else -80:]
t0=sel[0][0]
prev=t0
for s,e,n,qq,g,w in sel:
    print(f"{(s-t0)/1000:9.1f} gap {(s-prev)/1000:6.1f} dur {(e-s)/1000:6.1f} grid {int(g)//int(w):6d}x{w:4s} {n}")
    prev=e
