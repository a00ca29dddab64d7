"""Per hardware queue: how long a decode's kernel chain takes (first kernel's start to the end of
the packing kernel) and how long the queue then sits idle before the next chain starts.
Usage: chain_gaps.py <kernel_trace.csv> [first_chain [last_chain]] (chains counted per queue)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
lo = int(sys.argv[2]) if len(sys.argv) > 2 else 50
hi = int(sys.argv[3]) if len(sys.argv) > 3 else None
byq = collections.defaultdict(list)
for r in rows:
    name = r['Kernel_Name'].split('(')[0]
    if name.startswith('__amd'):
        continue
    byq[r['Queue_Id']].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), name))
for q, v in sorted(byq.items()):
    v.sort()
    chains, cur = [], []
    for k in v:
        cur.append(k)
        if k[2].endswith('k_pack') or k[2].endswith('k_parse_pack'):
            chains.append(cur)
            cur = []
    chains = chains[lo:hi]
    if len(chains) < 3:
        continue
    dur = sorted(c[-1][1] - c[0][0] for c in chains)
    busy = sorted(sum(k[1] - k[0] for k in c) for c in chains)
    idle = sorted(b[0][0] - a[-1][1] for a, b in zip(chains, chains[1:]))
    period = (chains[-1][0][0] - chains[0][0][0]) / (len(chains) - 1)
    print('queue %s: %d chains, %d kernels each; chain median %.1f us (kernels %.1f us), idle before the next chain median %.1f us, '
          'period %.1f us' % (q, len(chains), len(chains[0]), dur[len(dur) // 2] / 1e3, busy[len(busy) // 2] / 1e3,
                             idle[len(idle) // 2] / 1e3, period / 1e3))
