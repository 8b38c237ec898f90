"""Per-lane kernel timeline of one cnn_finetune step from a rocprofv3 kernel trace (CSV): start, duration, lane, name."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'decoder_fwd_persistent_kernel' in r['Kernel_Name']]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[which], idx[which + 1]
t0 = int(rows[a]['Start_Timestamp'])
keycol = 'Stream_Id' if 'Stream_Id' in rows[0] else 'Queue_Id'
lanes = {}
for r in rows[a:b]:
    ln = lanes.setdefault(r[keycol], len(lanes))
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
    g = r.get('Grid_Size', r.get('Grid_Size_X', ''))
    print('%9.1f %7.1f  %s%-52s grid %s' % ((s - t0) / 1e3, (e - s) / 1e3, '    ' * ln, n[:52], g))
