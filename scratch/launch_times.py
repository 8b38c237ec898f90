import csv, sys, glob
sys.path.insert(0,'.')
import comic_amd.nets as N
d = sys.argv[1]
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
names=[r['Kernel_Name'] for r in rows]
idx=[i for i,n in enumerate(names) if 'conv_stem' in n]
start=idx[-1]
plan=N.CnnPlan(group_branches=(len(sys.argv)<=2 or sys.argv[2]!='0'))
ops=plan.ops
launches=[]
i=0
while i<len(ops):
    o=ops[i]
    if o['kind'] in (5,6): i+=1; continue
    n=1
    if o.get('group',0)>0:
        while i+n<len(ops) and ops[i+n].get('group',0)==o['group']: n+=1
    launches.append(ops[i:i+n]); i+=n
seq=rows[start:start+len(launches)]
tot=0; gaps=0; prev_end=None; agg={}
for L,r in zip(launches,seq):
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    dt=(e-s)/1e3
    gap=(s-prev_end)/1e3 if prev_end else 0
    prev_end=e
    tot+=dt; gaps+=gap
    fl=sum(2*64*o['Ho']*o['Wo']*o['KH']*o['KW']*o['Cin']*o['Cout'] for o in L if o['kind']<2)
    kn=r['Kernel_Name']
    short=kn.split('(')[0].replace('void (anonymous namespace)::','').replace('unsigned short','bf16').replace('conv_igemm_dma_','')
    o=L[0]
    key=o['Ho']
    agg.setdefault(key,[0,0]); agg[key][0]+=dt+gap; agg[key][1]+=fl
    desc=' + '.join('%d>%d %dx%d'%(o['Cin'],o['Cout'],o['KH'],o['KW']) for o in L)
    print('%-36s %3dx%-3d %7.1f us gap %5.1f %6.1f TF/s wgs %5d  %s' % (short[:36],o['Ho'],o['Wo'],dt,gap,fl/dt/1e6 if dt else 0, int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])*int(r['Grid_Size_Y']), desc))
print('total kernel us',tot,'gaps',gaps, 'launches', len(launches))
for k,v in agg.items(): print('stage Ho=%d: %.1f us, %.1f TF/s'%(k,v[0],v[1]/v[0]/1e6))
