import csv,glob,re,sys
f=glob.glob(sys.argv[1]+'/trace/*/*kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
idx=[i for i,n in enumerate(names) if 'sr_head_fwd' in n]
a,b=idx[-2],idx[-1]
span=(int(rows[b]['Start_Timestamp'])-int(rows[a]['Start_Timestamp']))/1e3
busy=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows[a:b])/1e3
nas=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in rows[a:b] if 'nas_' in r['Kernel_Name'])/1e3
gaps=[(int(rows[i+1]['Start_Timestamp'])-int(rows[i]['End_Timestamp']))/1e3 for i in range(a,b)]
print('launches',b-a,'span us %.0f busy %.0f (nas body kernels %.0f) gaps sum %.0f median gap %.2f'%(span,busy,nas,sum(gaps),sorted(gaps)[len(gaps)//2]))
big=[(g,names[a+i][:50],names[a+i+1][:50]) for i,g in enumerate(gaps) if g>8]
for g,n1,n2 in sorted(big,reverse=True)[:12]: print('  gap %.1f us after %s -> %s'%(g,re.sub(r"<.*","",n1),re.sub(r"<.*","",n2)))
