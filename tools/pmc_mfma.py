"""Reduce the rocprofv3 --pmc passes of tools/pmc_mfma.sh to one JSON: per launch of ONE InceptionV3 forward the raw SQ / TCC /
GRBM counters, the launch duration of that pass, and the derived fractions.   pmc_mfma.py <dir> <B>

Derived (MI355X_MICROARCH.md, 'Per-instruction cycle constants' / 'rocprofv3 PMC slots'):
  mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (sum over XCDs)
  wait_frac / issue_stall_frac / active_frac = SQ_WAIT_ANY / SQ_WAIT_INST_ANY / SQ_ACTIVE_INST_ANY over SQ_WAVE_CYCLES
  lds_conflict_frac = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE;   l2_hit = TCC_HIT / (TCC_HIT + TCC_MISS)
"""
import csv, glob, json, os, sys

base, B = sys.argv[1], sys.argv[2]


def short(n):
    return n.replace('void (anonymous namespace)::', '').replace('(anonymous namespace)::', '').split('(')[0][:60]


def one_forward(d):
    fs = glob.glob(d + '/*/*counter_collection.csv')
    if not fs:
        return None
    rows = list(csv.DictReader(open(fs[0])))
    disp = {}
    for r in rows:
        k = int(r['Dispatch_Id'])
        e = disp.setdefault(k, dict(kernel=short(r['Kernel_Name']), grid=int(r['Grid_Size']), wg=int(r['Workgroup_Size']),
                                    lds=int(r['LDS_Block_Size']), vgpr=int(r['VGPR_Count']), agpr=int(r['Accum_VGPR_Count']),
                                    sgpr=int(r['SGPR_Count']),
                                    us=(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, c={}))
        e['c'][r['Counter_Name']] = e['c'].get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    ids = sorted(disp)
    first = 'conv_stem_mfma' if any('conv_stem_mfma' in disp[k]['kernel'] for k in ids) else 'conv_stem_stream'   # (kind 9: the stem conv is inside the streaming op)
    stems = [i for i, k in enumerate(ids) if first in disp[k]['kernel']]
    a, b = stems[-2], stems[-1]
    return [disp[k] for k in ids[a:b]]


sets = {}
for name in ('mfma', 'lds', 'l2'):
    s = one_forward('%s/pmc_%s_%s' % (base, name, B))
    if s is not None:
        sets[name] = s
n = {len(v) for v in sets.values()}
assert len(n) == 1, {k: len(v) for k, v in sets.items()}
launches = []
ref = next(iter(sets.values()))
for i in range(len(ref)):
    e = dict(kernel=ref[i]['kernel'], grid=ref[i]['grid'], workgroup=ref[i]['wg'], lds_bytes=ref[i]['lds'],
             vgpr=ref[i]['vgpr'], agpr=ref[i]['agpr'], sgpr=ref[i]['sgpr'])
    c = {}
    for name, s in sets.items():
        assert s[i]['kernel'] == e['kernel']
        c.update(s[i]['c'])
        e['us_pass_' + name] = round(s[i]['us'], 2)
    e['counters'] = c
    d = {}
    if c.get('GRBM_GUI_ACTIVE') and 'SQ_VALU_MFMA_BUSY_CYCLES' in c:
        cyc = c['GRBM_GUI_ACTIVE'] / 8.0
        d['kernel_cycles'] = cyc
        d['mfma_busy_frac'] = c['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024.0)
        if e.get('us_pass_mfma'):
            d['clock_ghz'] = cyc / (e['us_pass_mfma'] * 1e3)
    if c.get('SQ_WAVE_CYCLES'):
        for k, nm in (('SQ_WAIT_ANY', 'wait_frac'), ('SQ_WAIT_INST_ANY', 'issue_stall_frac'), ('SQ_ACTIVE_INST_ANY', 'active_frac')):
            if k in c:
                d[nm] = c[k] / c['SQ_WAVE_CYCLES']
    if c.get('SQ_LDS_IDX_ACTIVE'):
        d['lds_conflict_frac'] = c.get('SQ_LDS_BANK_CONFLICT', 0.0) / c['SQ_LDS_IDX_ACTIVE']
    if 'TCC_HIT_sum' in c:
        d['l2_hit'] = c['TCC_HIT_sum'] / max(c['TCC_HIT_sum'] + c.get('TCC_MISS_sum', 0.0), 1.0)
    e['derived'] = {k: round(v, 4) for k, v in d.items()}
    launches.append(e)
tot = {}
for e in launches:
    for k, v in e['counters'].items():
        tot[k] = tot.get(k, 0.0) + v
out = {
    'note': 'rocprofv3 --pmc passes (mfma / lds / l2 counter sets, each its own run with --kernel-trace only) over tools/run_cnn.py: '
            'one InceptionV3 forward (bf16, forward-only plan, autotuned tiles from a cache), eager launches, %s images. '
            'Durations are those of the profiled passes (counters slow the clock, never compare with un-profiled time).' % B,
    'images_per_forward': int(B), 'launches_per_forward': len(launches), 'totals': tot, 'per_launch': launches}
if tot.get('GRBM_GUI_ACTIVE') and 'SQ_VALU_MFMA_BUSY_CYCLES' in tot:
    out['forward_mfma_busy_frac'] = round(tot['SQ_VALU_MFMA_BUSY_CYCLES'] / (tot['GRBM_GUI_ACTIVE'] / 8.0 * 1024.0), 4)
json.dump(out, open('%s/cnn_mfma_counters_%s.json' % (base, B), 'w'), indent=1)
print('forward mfma busy frac', out.get('forward_mfma_busy_frac'), 'launches', len(launches))
for e in sorted(launches, key=lambda e: -e.get('us_pass_mfma', 0))[:12]:
    print('%-58s %8.1f us %s' % (e['kernel'], e.get('us_pass_mfma', 0), e['derived']))
