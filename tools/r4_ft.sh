#!/bin/bash
# cnn_finetune step: timing + kernel stats (rocprofv3 --kernel-trace --stats), one-step timeline
out=$GRAFT_REPO_ROOT/gpurun_out/r4_ft; mkdir -p $out
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export COMIC_TUNE_CACHE=$out/tiles.json
N=20 timeout -k 10 300 python3 tools/ft_step_time.py 2>&1 | tail -1
cd /tmp; rm -rf /tmp/kt
N=6 timeout -k 10 500 rocprofv3 --kernel-trace --stats -d /tmp/kt -o b --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/ft_step_time.py > $out/prof.log 2>&1 || { tail -20 $out/prof.log; exit 1; }
tail -1 $out/prof.log
cp /tmp/kt/b_kernel_stats.csv $out/ft_kernel_stats.csv
python3 - <<'P' > $out/ft_step_summary.txt
import csv
rows = list(csv.DictReader(open('/tmp/kt/b_kernel_trace.csv')))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'decoder_fwd_persistent_kernel' in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
t0 = int(rows[a]['Start_Timestamp']); t1 = int(rows[b]['Start_Timestamp'])
busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rows[a:b])
print('one finetune step: %.1f us wall, %.1f us kernel-busy, %d kernels' % ((t1 - t0) / 1e3, busy / 1e3, b - a))
agg = {}
for r in rows[a:b]:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]
    d = agg.setdefault(n, [0, 0]); d[0] += 1; d[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%8.1f us %4d x  %s' % (t / 1e3, c, n))
P
head -45 $out/ft_step_summary.txt
