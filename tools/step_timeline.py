"""Kernel sequence of one decoder training step from a rocprofv3 kernel trace (CSV): the dispatches between two
consecutive decoder_fwd_persistent_kernel launches, with start offsets, durations and the idle gap before each."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'decoder_fwd_persistent_kernel' in r['Kernel_Name']]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[which], idx[which + 1]
t0 = int(rows[a]['Start_Timestamp'])
prev_end = t0
busy = 0
for r in rows[a:b]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print('%9.1f %8.1f  gap %6.1f  %s' % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n[:90]))
    busy += e - s
    prev_end = max(prev_end, e)
print('step %.1f us, busy %.1f us, %d kernels' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3, busy / 1e3, b - a))
