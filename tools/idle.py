"""Chain-queue idle gaps > 60 us per step from a rocprofv3 kernel trace (development tool)."""
import csv, glob, re, sys
f = glob.glob(sys.argv[1] + '/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
fm = [i for i, r in enumerate(rows) if 'k_fill_masks' in r['Kernel_Name']]
starts = [fm[i] for i in range(len(fm)) if i == 0 or fm[i] - fm[i - 1] > 10]
short = lambda n: re.sub(r'\(.*', '', n).replace('void rau::', '').replace('rau::', '')[:40]
import collections
for si in range(3, min(6, len(starts) - 1)):
    step = rows[starts[si]:starts[si + 1]]
    t0 = step[0]['s']
    qs = collections.Counter(r['Queue_Id'] for r in step)
    cq = max(qs, key=qs.get)
    chain = [r for r in step if r['Queue_Id'] == cq]
    print('step', si, 'ms', (step[-1]['e'] - t0) / 1e6)
    prev = None
    for r in chain:
        if prev is not None and r['s'] - prev['e'] > 60000:
            print(f"   idle {(r['s']-prev['e'])/1e3:8.1f} us after {short(prev['Kernel_Name'])} @{(prev['e']-t0)/1e3:8.1f} before {short(r['Kernel_Name'])}")
        prev = r
