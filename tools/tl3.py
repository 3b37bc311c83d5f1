"""Phase summary of a RAU_PROF_TIMELINE dump: per stream first/last, phase marks on the chain."""
import sys
steps, cur = [], []
for line in open(sys.argv[1]):
    if line.startswith('#'):
        if cur: steps.append(cur)
        cur = []
        continue
    n, sid, a, b = line.strip().split(',')
    cur.append((n, int(sid), float(a), float(b)))
recs = steps[-1]
# two steps per dump: split at second fill_masks group
fm = [i for i, r in enumerate(recs) if r[0] == 'fill_masks']
if not fm:   # evaluate mode draws no masks: a step starts at the bulk stream's first transpose
    fm = [i for i, r in enumerate(recs) if r[0] == 'transpose']
starts = [fm[i] for i in range(len(fm)) if i == 0 or fm[i] - fm[i-1] > 10]
recs = recs[starts[-1]:]
t0 = min(r[2] for r in recs)
recs = [(n, s, a - t0, b - t0) for n, s, a, b in recs]
end = max(r[3] for r in recs)
print(f"step span {end:.3f} ms; kernels {len(recs)}")
for sid in (0, 1, 2):
    q = [r for r in recs if r[1] == sid]
    busy = sum(r[3] - r[2] for r in q)
    print(f"stream {sid}: n={len(q)} first {min(r[2] for r in q):.3f} last {max(r[3] for r in q):.3f} busy {busy:.3f}")
chain = sorted([r for r in recs if r[1] == 0], key=lambda r: r[2])
def mark(name, which=0):
    q = [r for r in chain if r[0] == name]
    return q[which] if q else None
for nm, wh in (('gather_q', 0), ('att_fwd_fused', 0), ('att_fwd_fused', -1), ('loss_reduce', 0), ('scale_hops', 0), ('att_bwd_fused', 0), ('att_bwd_fused', -1), ('dq_reduce', 0), ('embed_bwd', 0)):
    r = mark(nm, wh)
    if r: print(f"  chain {nm}[{wh}] {r[2]:.3f}-{r[3]:.3f}")
if len(sys.argv) > 2:
    for sid in (1, 2):
        for r in sorted([r for r in recs if r[1] == sid], key=lambda r: r[2]):
            print(f"  s{sid} {r[0]:18s} {r[2]:7.3f} {r[3]:7.3f} dur {1e3*(r[3]-r[2]):7.1f}")
    prev = None
    for r in chain:
        gap = (r[2] - prev[3]) * 1e3 if prev else 0
        print(f"  s0 {r[0]:18s} {r[2]:7.3f} {r[3]:7.3f} dur {1e3*(r[3]-r[2]):7.1f} gap {gap:6.1f}")
        prev = r
