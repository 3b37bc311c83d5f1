"""Summarise one step of a rocprofv3 kernel trace: bulk-stream kernels with gaps, and
chain-stream activity per window (development tool)."""
import collections
import csv
import glob
import re
import sys

f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp'])
    r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
fm = [i for i, r in enumerate(rows) if 'k_fill_masks' in r['Kernel_Name']]
starts = [fm[i] for i in range(len(fm)) if i == 0 or fm[i] - fm[i - 1] > 10]
a, b = starts[2], starts[3]
step = rows[a:b]
t0 = step[0]['s']


def short(n):
    n = re.sub(r'\(.*', '', n).replace('void rau::', '').replace('rau::', '')
    return n[:44]


qs = collections.Counter(r['Queue_Id'] for r in step)
chainq = max(qs, key=qs.get)
print('step ms', (max(r['e'] for r in step) - t0) / 1e6, dict(qs))
bulk = [r for r in step if r['Queue_Id'] != chainq]
chain = [r for r in step if r['Queue_Id'] == chainq]
prev = None
for r in bulk:
    gap = (r['s'] - prev) / 1e3 if prev else 0
    print(f"BULK {(r['s']-t0)/1e3:9.1f} dur {(r['e']-r['s'])/1e3:7.1f} gap {gap:7.1f} {short(r['Kernel_Name'])} grid {int(r['Grid_Size_X'])//256}")
    prev = r['e']
# chain busy fraction & markers
marks = [r for r in chain if 'att_fwd_fused' in r['Kernel_Name'] or 'att_bwd_fused' in r['Kernel_Name'] or 'k_gather_q' in r['Kernel_Name'] or 'k_embed_bwd' in r['Kernel_Name'] or 'k_dq_reduce' in r['Kernel_Name']]
for r in marks:
    print(f"CHAIN mark {(r['s']-t0)/1e3:9.1f} {short(r['Kernel_Name'])}")
print('chain end', (chain[-1]['e'] - t0) / 1e3, 'chain busy ms', sum(r['e'] - r['s'] for r in chain) / 1e6, 'n', len(chain))
