"""Chain-stream anatomy from a rocprofv3 --kernel-trace CSV (no HIP-event overhead in the gaps):
per phase of one step, kernel time and launch-to-launch gaps on the queue with the most dispatches.
usage: python tools/kt_chain.py <kernel_trace.csv> [step_index]"""
import collections, csv, re, sys
def short(n):
    n = n.replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("rau::", "")
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp']); r['n'] = short(r['Kernel_Name'])
rows.sort(key=lambda r: r['s'])
qn = collections.Counter(r['Queue_Id'] for r in rows)
chain_q = qn.most_common(1)[0][0]
# steps start at the first fill_masks-like kernel group: use k_embed_fwd as the step marker
marks = [i for i, r in enumerate(rows) if r['n'].startswith('k_embed_fwd')]
k = int(sys.argv[2]) if len(sys.argv) > 2 else len(marks) // 2
lo, hi = rows[marks[k]]['s'], rows[marks[k + 1]]['s']
step = [r for r in rows if lo <= r['s'] < hi]
print(f"step {k}: {(hi - lo) / 1e6:.3f} ms, {len(step)} kernels")
for q in qn:
    qr = [r for r in step if r['Queue_Id'] == q]
    if not qr: continue
    busy = sum(r['e'] - r['s'] for r in qr) / 1e6
    print(f" queue {q}: {len(qr)} kernels, busy {busy:.3f} ms, span {(qr[0]['s'] - lo) / 1e6:.3f}..{(qr[-1]['e'] - lo) / 1e6:.3f}")
ch = [r for r in step if r['Queue_Id'] == chain_q]
gaps = [ch[i + 1]['s'] - ch[i]['e'] for i in range(len(ch) - 1)]
print(f" chain: kernel time {sum(r['e'] - r['s'] for r in ch) / 1e6:.3f} ms, gaps {sum(gaps) / 1e6:.3f} ms "
      f"(median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us, n={len(gaps)})")
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for i, r in enumerate(ch):
    a = agg[r['n']]
    a[0] += 1; a[1] += (r['e'] - r['s']) / 1e3; a[2] += (gaps[i] if i < len(gaps) else 0) / 1e3
for n, a in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"   {n[:56]:56s} n={a[0]:3d} sum={a[1]:7.1f} us avg={a[1] / a[0]:6.1f} gap_after_avg={a[2] / a[0]:5.1f}")
