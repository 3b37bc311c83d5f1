"""Chain-stream idle gaps from a RAU_PROF_TIMELINE dump (HIP-event timeline, no rocprof)."""
import sys
steps, cur = [], []
for line in open(sys.argv[1]):
    if line.startswith('#'):
        if cur: steps.append(cur)
        cur = []
        continue
    n, sid, a, b = line.strip().split(',')
    cur.append((n, int(sid), float(a), float(b)))
for recs in steps[-1:]:
    # split into steps at fill_masks groups
    chain = sorted([r for r in recs if r[1] == 0], key=lambda r: r[2])
    prev = None
    for r in chain:
        if prev and r[2] - prev[3] > 0.06:
            print(f"idle {1e3*(r[2]-prev[3]):8.1f} us after {prev[0]} @{prev[3]:8.3f} before {r[0]}")
        prev = r
    for sid in (1, 2):
        q = sorted([r for r in recs if r[1] == sid], key=lambda r: r[2])
        print('stream', sid, 'first', q[0][2], 'last', q[-1][3], 'n', len(q))
    print('chain span', chain[0][2], chain[-1][3])
