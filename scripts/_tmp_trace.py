import csv,glob,sys
d=sys.argv[1]
ev=[]
for r in csv.DictReader(open(glob.glob(d+'/*kernel_trace.csv')[0])):
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'q'+r['Queue_Id']+' '+r['Kernel_Name'].split('(')[0][-52:]))
for r in csv.DictReader(open(glob.glob(d+'/*memory_copy_trace.csv')[0])):
    ev.append((int(r['Start_Timestamp']),int(r['End_Timestamp']),'COPY '+r['Direction'][12:]))
ev.sort()
end=ev[-1][0]
sel=[e for e in ev if e[0] > end-int(float(sys.argv[2])*1e9) and e[0] < end-int(float(sys.argv[3])*1e9)]
t0=sel[0][0]
for s,e,n in sel:
    if (e-s) > 15000 or n.startswith('COPY'): print(f"{(s-t0)/1000:9.1f} dur {(e-s)/1000:8.1f} {n}")
print('events', len(sel))
