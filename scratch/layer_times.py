import csv, sys, glob
sys.path.insert(0,'.')
import comic_amd.nets as N
d = sys.argv[1]
f = glob.glob(d + '/*/*_kernel_trace.csv')[0]
rows=list(csv.DictReader(open(f)))
names=[r['Kernel_Name'] for r in rows]
idx=[i for i,n in enumerate(names) if 'conv_stem' in n]
start=idx[-1]
plan=N.CnnPlan()
ops=plan.ops
seq=rows[start:start+len(ops)]
tot=0; agg={}
for o,r in zip(ops,seq):
    dt=(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3
    tot+=dt
    M=64*o['Ho']*o['Wo']
    fl=2*M*o['KH']*o['KW']*o['Cin']*o['Cout'] if o['kind']<2 else 0
    kn=r['Kernel_Name']
    short=kn.split('(')[0].replace('void (anonymous namespace)::','').replace('unsigned short','bf16')
    key=(o['Ho'],)
    agg.setdefault(key,[0,0]); agg[key][0]+=dt; agg[key][1]+=fl
    if len(sys.argv)>2:
        print('%-44s %3dx%-3d Cin%4d Cout%4d %dx%d s%d M=%6d %7.1f us %6.1f TF/s grid %sx%s' % (short[:44],o['Ho'],o['Wo'],o['Cin'],o['Cout'],o['KH'],o['KW'],o['SH'],M,dt,fl/dt/1e6 if dt else 0, int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']), r['Grid_Size_Y']))
print('total us',tot)
for k,v in agg.items(): print('stage Ho=%d: %.1f us, %.1f TF/s'%(k[0],v[0],v[1]/v[0]/1e6))
