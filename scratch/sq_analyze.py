import csv, glob, collections, sys
sys.path.insert(0,'.')
import comic_amd.nets as N
plan=N.CnnPlan()
ops=[o for o in plan.ops if o['kind'] not in (5,6)]
f=glob.glob(sys.argv[1]+'/*/*counter_collection.csv')[0]
rows=list(csv.DictReader(open(f)))
disp=collections.OrderedDict()
for r in rows:
    d=disp.setdefault(r['Dispatch_Id'],{'name':r['Kernel_Name'],'t':(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3,'grid':r['Grid_Size'],'wg':r['Workgroup_Size']})
    d[r['Counter_Name']]=float(r['Counter_Value'])
dl=list(disp.values())
idx=[i for i,d in enumerate(dl) if 'conv_stem' in d['name']]
seq=dl[idx[-1]:idx[-1]+len(ops)]
print('%-46s %7s %6s | %8s %6s %6s %6s | %7s %7s %6s %7s'%('layer','us','waves','cyc/wave','wait%','winst%','act%','VALU/w','SALU/w','bankcf%','mfma%'))
for o,d in zip(ops,seq):
    if o['kind']!=0: continue
    waves=int(d['grid'])/64
    wc=d['SQ_WAVE_CYCLES']
    print('%3dx%-3d Cin%4d Cout%4d %dx%d %-16s %7.1f %6d | %8.0f %6.1f %6.1f %6.1f | %7.0f %7.0f %6.1f %7.1f'%(o['Ho'],o['Wo'],o['Cin'],o['Cout'],o['KH'],o['KW'],d['name'].split('<')[1].split('>')[0][:16],d['t'],waves,wc/waves*4, 100*d['SQ_WAIT_ANY']/wc,100*d['SQ_WAIT_INST_ANY']/wc,100*d['SQ_ACTIVE_INST_ANY']/wc,d['SQ_INSTS_VALU']/waves,d['SQ_INSTS_SALU']/waves,100*d['SQ_LDS_BANK_CONFLICT']/max(1,wc*4), 100*d['SQ_VALU_MFMA_BUSY_CYCLES']/(wc*4)))
