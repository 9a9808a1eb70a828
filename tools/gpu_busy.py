#!/usr/bin/env python3
"""GPU-busy fraction of a steady-state window of a `rocprofv3 --kernel-trace` run.

  python tools/gpu_busy.py <kernel_trace.csv> [--skip 0.3] [--marker adam_kernel]

Busy = length of the UNION of the kernel intervals (kernels of two streams overlap: a plain sum of durations would count
the chip twice) over the wall time of the window; `sum` is the plain sum beside it.  The window is the last (1 - skip) of
the run cut at step boundaries: a step ends with the marker kernel (the optimizer launch of the training step), so the
window starts right after one marker and ends with the last one -- whole steps only, no warm-up, no fence.  The largest
gaps inside the window are listed (a gap above ~5 us is a launch the host delivered late)."""
import argparse
import csv
import json
import sys


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--skip', type=float, default=0.3, help='fraction of the markers skipped at the start (warm-up)')
    ap.add_argument('--marker', default='adam_kernel')
    ap.add_argument('--last', type=int, default=0, help='use exactly the last N steps (0: by --skip)')
    args = ap.parse_args()
    rows = []
    with open(args.trace) as f:
        for r in csv.DictReader(f):
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
    rows.sort()
    marks = [i for i, r in enumerate(rows) if args.marker in r[2]]
    if len(marks) < 3:
        print(json.dumps({'error': 'fewer than 3 marker kernels (%s) in the trace' % args.marker}))
        return 1
    first = len(marks) - 1 - args.last if args.last else int(len(marks) * args.skip)
    first = max(0, min(first, len(marks) - 2))
    lo, hi = marks[first] + 1, marks[-1]
    steps = len(marks) - 1 - first
    win = rows[lo:hi + 1]
    t0, t1 = rows[marks[first]][1], win[-1][1]
    busy = 0
    cur_s, cur_e = None, None
    gaps = []
    prev_end = t0
    for s, e, name in win:
        s = max(s, t0)
        if cur_e is None:
            cur_s, cur_e = s, e
        elif s <= cur_e:
            cur_e = max(cur_e, e)
        else:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        if s > prev_end:
            gaps.append((s - prev_end, name))
        prev_end = max(prev_end, e)
    busy += cur_e - cur_s
    total = t1 - t0
    ksum = sum(e - s for s, e, _ in win)
    gaps.sort(reverse=True)
    out = {'steps': steps, 'kernels_per_step': round(len(win) / steps, 1), 'ms_per_step': round(total / steps / 1e6, 4),
           'busy_ms_per_step': round(busy / steps / 1e6, 4), 'kernel_sum_ms_per_step': round(ksum / steps / 1e6, 4),
           'gpu_busy_frac': round(busy / total, 4), 'idle_ms_per_step': round((total - busy) / steps / 1e6, 4),
           'gaps_over_5us_per_step': round(sum(1 for g, _ in gaps if g > 5000) / steps, 1),
           'largest_gaps_us': [[round(g / 1e3, 1), n[:60]] for g, n in gaps[:8]]}
    print(json.dumps(out))
    return 0


if __name__ == '__main__':
    sys.exit(main())
