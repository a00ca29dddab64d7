"""gpurun_out/profile_<workload> (written by tools/profile.sh on the GPU box) -> profiles/.

    python tools/summarize_profile.py <tag> <workload>

Copies the kernel-trace statistics and the two bench lines (plain and under rocprofv3) and
condenses the FETCH_SIZE / WRITE_SIZE passes into profiles/pmc_<workload>.json, which bench.py
reads for the `traffic` fields of its roofline object."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DST = os.path.join(ROOT, 'profiles')
tag = sys.argv[1] if len(sys.argv) > 1 else 'r2'
workload = sys.argv[2] if len(sys.argv) > 2 else 'b64_368'
SRC = os.path.join(ROOT, 'gpurun_out', 'profile_' + workload)


def newest(pattern):
    # gpurun merges into gpurun_out/ without removing the files of earlier calls
    return max(glob.glob(pattern), key=os.path.getmtime)


def counters(kind):
    f = newest(os.path.join(SRC, kind, '*', '*counter_collection.csv'))
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    # mean over all launches of a kernel: launches of one kernel can differ in size (the resize
    # kernel runs once for the 19 heatmaps and once for the 38 PAF channels of a step)
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


os.makedirs(DST, exist_ok=True)
shutil.copy(newest(os.path.join(SRC, 'trace', '*', '*kernel_stats.csv')),
            os.path.join(DST, '%s_%s_kernel_stats.csv' % (tag, workload)))
shutil.copy(os.path.join(SRC, 'bench.json'), os.path.join(DST, '%s_%s_bench.json' % (tag, workload)))
shutil.copy(os.path.join(SRC, 'trace_bench.json'), os.path.join(DST, '%s_%s_bench_under_rocprof.json' % (tag, workload)))
fetch, write = counters('fetch'), counters('write')
# every decode ends with one packing launch (k_parse_pack, or k_pack where assembly and packing are apart)
decodes = max([n for k, (v, n) in list(fetch.items()) + list(write.items()) if k.startswith('k_pack') or k.startswith('k_parse_pack')] + [1])
out = {'workload': workload, 'unit': 'bytes per launch', 'build': tag,
       'method': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs of "bench.py --workload %s --no-cpu '
                 '--no-extras --steps 6 --warmup 2 --contexts 1"; mean over the launches of a kernel; counters are in '
                 'KiB; FETCH_SIZE is doubled (gfx950 tallies 128-byte read requests at 64 bytes; checked on '
                 'k_heat_planar, whose coalesced read of B*h*w*19*4 bytes is reported at one half)' % workload,
       'kernels': {}}
step = 0.0
resize = 0.0
for k in sorted(set(fetch) | set(write)):
    f, nf = fetch.get(k, (0.0, 0))
    w, nw = write.get(k, (0.0, 0))
    f *= 1024.0
    w *= 1024.0
    name = k.split('(')[0].replace('void ', '')
    per_step = max(nf, nw) / float(max(decodes, 1))
    out['kernels'][name] = {'fetch_raw': f, 'fetch_corrected': 2.0 * f, 'write': w, 'hbm_bytes': 2.0 * f + w,
                            'launches_per_decode': round(per_step, 2)}
    if name.startswith('k_') and per_step >= 0.5:
        step += (2.0 * f + w) * round(per_step)
    if 'resize' in name:
        resize += (2.0 * f + w) * round(per_step)
out['step_hbm_bytes'] = step
try:    # images per decode of the run the counters were read on (bench.py scales by it for other chunk sizes)
    with open(os.path.join(SRC, 'fetch_bench.json')) as fh:
        cfg = json.loads(fh.read().strip().splitlines()[-1])['config']
    out['batch'] = cfg.get('images_per_decode', cfg.get('batch_per_gpu'))
except Exception:
    pass
if resize:
    out['resize_stage_hbm_bytes'] = resize
with open(os.path.join(DST, 'pmc_%s.json' % workload), 'w') as fh:
    json.dump(out, fh, indent=1)
print(json.dumps({k: (round(v['hbm_bytes'] / 1e6, 3), v['launches_per_decode']) for k, v in out['kernels'].items()}, indent=1))
print('step', step / 1e6, 'MB; resize stage', resize / 1e6, 'MB')
