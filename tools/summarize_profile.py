"""gpurun_out/profile (written by tools/profile.sh on the GPU box) -> profiles/."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, 'gpurun_out', 'profile')
DST = os.path.join(ROOT, 'profiles')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r1'
workload = 'b64_368'


def newest(pattern):
    # gpurun merges into gpurun_out/ without removing the files of earlier calls
    return max(glob.glob(pattern), key=os.path.getmtime)


def counters(kind):
    f = newest(os.path.join(SRC, kind, '*', '*counter_collection.csv'))
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    # the first launches run eagerly with cold caches; the steady state is the median
    return {k: sorted(v)[len(v) // 2] for k, v in agg.items()}


os.makedirs(DST, exist_ok=True)
shutil.copy(newest(os.path.join(SRC, 'trace', '*', '*kernel_stats.csv')),
            os.path.join(DST, '%s_kernel_stats.csv' % tag))
shutil.copy(os.path.join(SRC, 'bench.json'), os.path.join(DST, '%s_bench.json' % tag))
shutil.copy(os.path.join(SRC, 'trace_bench.json'), os.path.join(DST, '%s_bench_under_rocprof.json' % tag))
fetch, write = counters('fetch'), counters('write')
out = {'workload': workload, 'unit': 'bytes per launch',
       'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of '
                 '"bench.py --no-cpu --steps 12 --warmup 3 --contexts 1"; counters are in KiB; FETCH_SIZE is doubled '
                 '(gfx950 tallies 128-byte read requests at 64 bytes; checked here on k_heat_planar, whose '
                 'coalesced read of B*h*w*19*4 bytes is reported at one half)',
       'kernels': {}}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0) * 1024.0, write.get(k, 0.0) * 1024.0
    out['kernels'][k] = {'fetch_raw': f, 'fetch_corrected': 2.0 * f, 'write': w, 'hbm_bytes': 2.0 * f + w}
for k, v in out['kernels'].items():
    if 'k_cell_filter' in k:
        out['roofline_kernel'] = k
        out['roofline_kernel_hbm_bytes_per_launch'] = v['hbm_bytes']
with open(os.path.join(DST, 'pmc_%s.json' % workload), 'w') as fh:
    json.dump(out, fh, indent=1)
print(json.dumps({k: round(v['hbm_bytes'] / 1e6, 3) for k, v in out['kernels'].items()}, indent=1))
