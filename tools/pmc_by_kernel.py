"""Reduce rocprofv3 --pmc passes (mfma / lds / l2 sets of tools/pmc_any.sh) to one line per (kernel, grid): mean duration and
the derived fractions of tools/pmc_mfma.py.   pmc_by_kernel.py <dir> [name-filter]"""
import csv, glob, sys, json
base = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ''
agg = {}
for name in ('mfma', 'lds', 'l2'):
    fs = glob.glob('%s/pmc_%s/*/*counter_collection.csv' % (base, name))
    if not fs:
        continue
    disp = {}
    for r in csv.DictReader(open(fs[0])):
        if flt and flt not in r['Kernel_Name']:
            continue
        k = int(r['Dispatch_Id'])
        e = disp.setdefault(k, dict(key=(r['Kernel_Name'].replace('void (anonymous namespace)::', '').split('(')[0][:48], int(r['Grid_Size'])),
                                    us=(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, c={}))
        e['c'][r['Counter_Name']] = e['c'].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    for e in disp.values():
        a = agg.setdefault(e['key'], {})
        s = a.setdefault(name, dict(n=0, us=0.0, c={}))
        s['n'] += 1; s['us'] += e['us']
        for k, v in e['c'].items():
            s['c'][k] = s['c'].get(k, 0.0) + v
out = []
for key, a in sorted(agg.items(), key=lambda kv: -kv[1].get('mfma', {'us': 0})['us']):
    c, d = {}, {}
    for name, s in a.items():
        for k, v in s['c'].items():
            c[k] = v / s['n']
        d['us_' + name] = round(s['us'] / s['n'], 1)
    if c.get('GRBM_GUI_ACTIVE'):
        cyc = c['GRBM_GUI_ACTIVE'] / 8.0
        d['mfma_busy'] = round(c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (cyc * 1024.0), 3)
        d['ghz'] = round(cyc / (d['us_mfma'] * 1e3), 2)
    if c.get('SQ_WAVE_CYCLES'):
        for k, nm in (('SQ_WAIT_ANY', 'wait'), ('SQ_WAIT_INST_ANY', 'stall'), ('SQ_ACTIVE_INST_ANY', 'active')):
            if k in c:
                d[nm] = round(c[k] / c['SQ_WAVE_CYCLES'], 3)
        d['occ_waves_per_simd'] = round(c['SQ_WAVE_CYCLES'] * 4 / max(c.get('GRBM_GUI_ACTIVE', 1) / 8.0 * 1024.0, 1), 2)
    if c.get('SQ_LDS_IDX_ACTIVE'):
        d['lds_conflict'] = round(c.get('SQ_LDS_BANK_CONFLICT', 0.0) / c['SQ_LDS_IDX_ACTIVE'], 3)
    if 'TCC_HIT_sum' in c:
        d['l2_hit'] = round(c['TCC_HIT_sum'] / max(c['TCC_HIT_sum'] + c.get('TCC_MISS_sum', 0.0), 1.0), 3)
        d['l2_req_M'] = round((c['TCC_HIT_sum'] + c.get('TCC_MISS_sum', 0.0)) / 1e6, 2)
    if 'SQ_INSTS_VALU' in c:
        d['valu_insts_M'] = round(c['SQ_INSTS_VALU'] / 1e6, 2)
    print('%-50s grid %8d  %s' % (key[0], key[1], d))
    out.append(dict(kernel=key[0], grid=key[1], derived=d, counters=c))
json.dump(out, open(base + '/pmc_by_kernel.json', 'w'), indent=1)
