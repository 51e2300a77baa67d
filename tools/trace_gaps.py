"""Look-ahead frame: is the main stream device-bound or waiting (for launches / for the side stream)?  From a rocprofv3 --kernel-trace CSV
of bench.py: frames are cut at the sensory update's first launch (AREA_DOWN3: one per frame, caller's queue; a 'frame' therefore runs from
the tail of frame t -- sensory update, memorising -- to the decoder of frame t + 1); for the frames [f0, f1) the main queue's busy time, its
idle gaps by the kernel that FOLLOWS the gap, and the side queue's busy time per frame.
    python tools/trace_gaps.py <kernel_trace.csv> f0 f1 [--dump f]      (--dump f: every launch of frame f on the main queue: gap before, duration)"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
f0, f1 = int(sys.argv[2]), int(sys.argv[3])
q = defaultdict(list)
for r in rows:
    q[r['Queue_Id']].append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:48]))
# frame marker: a kernel that runs exactly once per propagated frame on the caller's queue.  (Round 3 used query_init2_kernel; since round 4
# QUERY_INIT runs only when the object summaries changed, so the decoder's last launch -- up4_softmax -- marks the frames.)
MARK = 'area_down3'                                       # (round 5: up4_softmax runs on the auxiliary stream when the caller gives hints;
                                                          # AREA_DOWN3 opens the sensory update of every frame on the caller's queue)
main = max(q, key=lambda k: sum(MARK in n for _, _, n in q[k]))
side = [k for k in q if k != main]
for v in q.values():
    v.sort()
mq = q[main]
marks = [s for s, e, n in mq if MARK in n]
print('queues', {k: len(v) for k, v in q.items()}, 'main', main, 'frames', len(marks))
t0, t1 = marks[f0], marks[f1]
nf = f1 - f0
win = [x for x in mq if t0 <= x[0] < t1]
busy = sum(e - s for s, e, _ in win)
print(f'frames {f0}..{f1}: {(t1 - t0) / nf / 1e3:.1f} us per frame; main queue busy {busy / nf / 1e3:.1f} us per frame ({busy / (t1 - t0) * 100:.1f} %), {len(win) / nf:.1f} launches per frame')
gap_by = defaultdict(lambda: [0, 0])
for a, b in zip(win[:-1], win[1:]):
    g = b[0] - a[1]
    if g > 1500:
        gap_by[b[2]][0] += g; gap_by[b[2]][1] += 1
tot = sum(v[0] for v in gap_by.values())
print(f'idle in gaps > 1.5 us: {tot / nf / 1e3:.1f} us per frame; by the kernel after the gap:')
for n, (g, c) in sorted(gap_by.items(), key=lambda kv: -kv[1][0])[:14]:
    print(f'   {n:50s} {g / nf / 1e3:7.1f} us per frame in {c / nf:5.2f} gaps per frame (mean {g / c / 1e3:.1f} us)')
small = [b[0] - a[1] for a, b in zip(win[:-1], win[1:]) if 0 < b[0] - a[1] <= 1500]
print(f'gaps <= 1.5 us: {len(small) / nf:.1f} per frame, mean {sum(small) / max(1, len(small)) / 1e3:.2f} us, total {sum(small) / nf / 1e3:.1f} us per frame')
for k in side:
    w = [x for x in q[k] if t0 <= x[0] < t1]
    if w:
        print(f'queue {k}: {len(w) / nf:.1f} launches per frame, busy {sum(e - s for s, e, _ in w) / nf / 1e3:.1f} us per frame')

if '--dump' in sys.argv:
    f = int(sys.argv[sys.argv.index('--dump') + 1])
    a, b = marks[f], marks[f + 1]
    win = [x for x in mq if a <= x[0] < b]
    prev = None
    print(f'--- main queue, frame {f}: {(b - a) / 1e3:.1f} us, {len(win)} launches (gap before | duration | kernel)')
    agg = defaultdict(lambda: [0, 0, 0])
    for s_, e_, n_ in win:
        g = 0 if prev is None else s_ - prev
        print(f'  {g / 1e3:7.2f} {(e_ - s_) / 1e3:8.2f}  {n_}')
        agg[n_][0] += 1; agg[n_][1] += e_ - s_; agg[n_][2] += max(g, 0)
        prev = e_
    print('--- by kernel: launches, busy us, gaps-before us')
    for n_, (c, d, g) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f'  {c:3d} {d / 1e3:8.1f} {g / 1e3:7.1f}  {n_}')
    for k in side:
        w = [x for x in q[k] if a <= x[0] < b]
        if w:
            print(f'--- queue {k} during frame {f}: {len(w)} launches, busy {sum(e - s for s, e, _ in w) / 1e3:.1f} us, first at +{(w[0][0] - a) / 1e3:.1f} us, last ends +{(w[-1][1] - a) / 1e3:.1f} us')
