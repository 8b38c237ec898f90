"""Lanes of one cnn_finetune step from a rocprofv3 kernel trace (CSV): per stream / queue the busy time, the kernel count and
the idle gaps inside the step (between two consecutive decoder_fwd_persistent_kernel launches), then the phases of the step
on the union of the lanes (time during which 0 / 1 / 2 / 3+ kernels run)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'decoder_fwd_persistent_kernel' in r['Kernel_Name']]
which = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2
a, b = idx[which], idx[which + 1]
t0, t1 = int(rows[a]['Start_Timestamp']), int(rows[b]['Start_Timestamp'])
keycol = 'Stream_Id' if 'Stream_Id' in rows[0] else 'Queue_Id'
step = [r for r in rows if t0 <= int(r['Start_Timestamp']) < t1]
lanes = collections.defaultdict(list)
for r in step:
    lanes[r[keycol]].append(r)
print('step %.1f us, %d kernels, lanes by %s' % ((t1 - t0) / 1e3, len(step), keycol))
for k, rs in sorted(lanes.items(), key=lambda kv: -len(kv[1])):
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs)
    first, last = int(rs[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in rs)
    names = collections.Counter(r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:40] for r in rs)
    print('lane %s: %4d kernels, busy %7.1f us, span %7.1f..%7.1f us; %s' % (k, len(rs), busy / 1e3, (first - t0) / 1e3, (last - t0) / 1e3,
                                                                           ', '.join('%s x%d' % kv for kv in names.most_common(4))))
ev = []
for r in step:
    ev.append((int(r['Start_Timestamp']), 1)); ev.append((int(r['End_Timestamp']), -1))
ev.sort()
depth, prev, hist = 0, t0, collections.Counter()
for t, d in ev:
    hist[min(depth, 3)] += t - prev
    prev, depth = t, depth + d
print('kernels in flight: ' + ', '.join('%d%s: %.1f us' % (k, '+' if k == 3 else '', v / 1e3) for k, v in sorted(hist.items())))
# the twenty longest idle stretches of the whole device
ev2 = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in step)
end = t0; gaps = []
for s, e, n in ev2:
    if s > end: gaps.append((s - end, (end - t0) / 1e3, n.replace('(anonymous namespace)::', '')[:60]))
    end = max(end, e)
gaps.sort(reverse=True)
print('device idle in total %.1f us; longest:' % (sum(g[0] for g in gaps) / 1e3))
for g in gaps[:12]:
    print('  %.1f us at %.1f before %s' % (g[0] / 1e3, g[1], g[2]))
