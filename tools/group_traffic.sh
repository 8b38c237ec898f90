#!/bin/bash
# HBM-side reads and L2 hit rate of one grouped launch per tile id: group_traffic.sh <out> (env OP, TILES, B as tools/group_run.py)
out=$1; mkdir -p $out; export TMPDIR=/tmp
for c in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  name=$(echo $c | cut -d' ' -f1)
  rm -rf $out/pmc_$name
  timeout -k 10 300 rocprofv3 --pmc $c --kernel-trace -d $out/pmc_$name --output-format csv -- python3 tools/group_run.py > $out/pmc_$name.log 2>&1 || { echo "pass $name failed"; tail -3 $out/pmc_$name.log; }
done
python3 - $out <<'PY'
import csv, glob, sys
out = sys.argv[1]
for name in ('FETCH_SIZE', 'TCC_HIT_sum'):
    f = glob.glob('%s/pmc_%s/*/*counter_collection.csv' % (out, name))[0]
    agg = {}
    for r in csv.DictReader(open(f)):
        if 'conv_igemm' not in r['Kernel_Name']: continue
        k = (r['Kernel_Name'].split('::')[-1].split('(')[0][:60], r['Grid_Size'])
        d = agg.setdefault(k, {}).setdefault(r['Dispatch_Id'], {})
        d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        d['us'] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    for k, ds in agg.items():
        ds = list(ds.values())[-4:]
        m = {c: sum(d.get(c, 0) for d in ds) / len(ds) for c in ds[0]}
        if name == 'FETCH_SIZE':
            print('%-62s grid %-8s %7.1f us  read %.0f MB' % (k[0], k[1], m['us'], m['FETCH_SIZE'] * 2 * 1024 / 1e6))
        else:
            print('%-62s grid %-8s %7.1f us  L2 hit %.3f' % (k[0], k[1], m['us'], m['TCC_HIT_sum'] / max(m['TCC_HIT_sum'] + m['TCC_MISS_sum'], 1)))
PY
