import csv, glob, sys, collections, re
d = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else 'conv'
for f in sorted(glob.glob(d + '/p*/**/*counter_collection.csv', recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        m = re.search(r'(\w*%s\w*(<[^>]*>)?)' % pat, r['Kernel_Name'])
        if not m: continue
        agg[m.group(1)][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, c in agg.items():
        print(k)
        for n, v in c.items():
            print('   %-28s %14.0f  (n=%d)' % (n, sum(v[2:]) / max(len(v[2:]), 1), len(v)))
