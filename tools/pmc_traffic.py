"""profiles/<round>_cnn_hbm_traffic.json from two rocprofv3 --pmc passes over tools/run_cnn.py:  pmc_traffic.py <dir> <out.json>"""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
def per_forward(d, counter):
    f = glob.glob(d + '/*/*counter_collection.csv')[0]
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    first = 'conv_stem_mfma' if any('conv_stem_mfma' in r['Kernel_Name'] for r in rows) else 'conv_stem_stream'
    stems = [i for i, r in enumerate(rows) if first in r['Kernel_Name']]   # first launch of a forward
    a, b = stems[-2], stems[-1]            # one whole forward: stem conv .. next stem conv
    seg = rows[a:b]
    conv = sum(float(r['Counter_Value']) for r in seg if 'conv_' in r['Kernel_Name'] and 'pack_conv' not in r['Kernel_Name'])
    allk = sum(float(r['Counter_Value']) for r in seg)
    return conv, allk, len(seg), [(r['Kernel_Name'].split('(')[0].replace('void (anonymous namespace)::', '')[:50], float(r['Counter_Value'])) for r in seg]
base = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out'
fc, fa, n, fl = per_forward(base + '/pmc_FETCH_SIZE', 'FETCH_SIZE')
wc, wa, n2, wl = per_forward(base + '/pmc_WRITE_SIZE', 'WRITE_SIZE')
assert n == n2, (n, n2)
out = {
 'note': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, eager launches, batch %s, bf16' % os.environ.get('B', '640') + ', grouped forward-only plan (pool-after-projection + fuse_pools rewrites), autotuned tiles from a cache). Per MI355X_MICROARCH.md HBM section: counters are in KB (x1024 bytes); on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads, so read bytes = 2 x FETCH_SIZE x 1024; WRITE_SIZE is exact. Infinity-Cache hits are counted.',
 'command': 'tools/prof_bench.sh <dir>: COMIC_TUNE_CACHE=<dir>/tiles.json rocprofv3 --pmc <C> --kernel-trace --output-format csv -- python3 tools/run_cnn.py   (tools/pmc_traffic.py <dir> reduces the two counter_collection.csv files)',
 'launches_per_forward': n,
 'images_per_forward': int(os.environ.get('B', '640')),
 'per_forward': {'fetch_size_kb': fa, 'write_size_kb': wa, 'hbm_bytes_corrected': (2 * fa + wa) * 1024,
                 'conv_only_bytes_corrected': (2 * fc + wc) * 1024},
 'per_launch': [dict(kernel=k, FETCH_SIZE_KB=v, WRITE_SIZE_KB=w[1]) for (k, v), w in zip(fl, wl)],
}
json.dump(out, open(sys.argv[2] if len(sys.argv) > 2 else 'profiles/cnn_hbm_traffic.json', 'w'), indent=1)
print(json.dumps(out['per_forward']), n)
