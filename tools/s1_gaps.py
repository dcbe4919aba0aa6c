"""Idle time of the GPU inside one Stage-1 step, from tools/trace_stage1.sh's per-dispatch trace (queue, start_ns, end_ns, kernel).

    python tools/s1_gaps.py gpurun_out/<tag>_stage1_trace.csv

Prints the union-busy time against the wall time of the last full step (adam_clip_kernel to adam_clip_kernel), the busy time per queue, and the largest gaps
(nothing running on any queue) with the kernels on either side."""
import csv
import sys


def main():
    rows = [(r['queue'], int(r['start_ns']), int(r['end_ns']), r['kernel']) for r in csv.DictReader(open(sys.argv[1]))]
    rows.sort(key=lambda r: r[1])
    ends = [i for i, r in enumerate(rows) if r[3].startswith('adam_clip_kernel')]
    if len(ends) >= 2:
        rows = rows[ends[-2] + 1:ends[-1] + 1]
    t0, t1 = rows[0][1], max(r[2] for r in rows)
    busy, cur_end, gaps = 0, t0, []
    last = None
    for q, s, e, k in rows:
        if s > cur_end:
            gaps.append((s - cur_end, last, k, cur_end - t0))
            busy += 0
            cur_start = s
        busy += max(0, e - max(s, cur_end))
        if e > cur_end:
            cur_end, last = e, k
    print(f'step wall {(t1 - t0) / 1e6:.2f} ms, some kernel running {busy / 1e6:.2f} ms, idle {(t1 - t0 - busy) / 1e6:.2f} ms in {len(gaps)} gaps; {len(rows)} dispatches')
    per_q = {}
    for q, s, e, k in rows:
        per_q[q] = per_q.get(q, 0) + e - s
    for q, v in sorted(per_q.items(), key=lambda x: -x[1]):
        print(f'  queue {q}: {v / 1e6:.2f} ms of kernels')
    hist = {}
    for g, a, b, at in gaps:
        key = '<2us' if g < 2000 else '<5us' if g < 5000 else '<10us' if g < 10000 else '<50us' if g < 50000 else '>=50us'
        c = hist.setdefault(key, [0, 0])
        c[0] += 1
        c[1] += g
    for k_, (n, tot) in hist.items():
        print(f'  gaps {k_}: {n} totalling {tot / 1e6:.2f} ms')
    print('largest gaps:')
    for g, a, b, at in sorted(gaps, key=lambda x: -x[0])[:15]:
        print(f'  {g / 1e3:8.1f} us at {at / 1e6:6.2f} ms  after {str(a)[:50]}  before {b[:50]}')


if __name__ == '__main__':
    main()
