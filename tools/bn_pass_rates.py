"""Per-launch durations of the batch-norm passes of the training step, by position inside the step.

    python tools/bn_pass_rates.py <kernel_trace.csv> [--per-step 13] [--names bn_reduce_kernel<2>,bn_bwd_apply_fast]

The kernel-trace CSV of `rocprofv3 --kernel-trace` lists launches in dispatch order; the k-th launch of a kernel inside a
step is always the same layer (backward visits conv5_3 .. conv1_1), so the median duration per (kernel, k mod per_step)
is that layer's duration.  Run with XV_WGRAD_STREAM=0 (one stream): a kernel's duration is then its own."""
import argparse
import collections
import csv
import statistics


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--names', default='bn_reduce_kernel<2>,bn_bwd_apply_fast_kernel,bn_pool_bwd_kernel<0>,bn_pool_bwd_kernel<1>,'
                    'bn_apply_fast_kernel,bn_apply_pool_kernel,bn_reduce_kernel<0>,bn_rows_kernel')
    args = ap.parse_args()
    calls = collections.defaultdict(list)
    with open(args.trace) as f:
        for r in csv.DictReader(f):
            nm = r['Kernel_Name']
            for key in args.names.split(','):
                if key in nm:
                    calls[key].append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp']),
                                       int(r['Grid_Size_X'])))
    for key, lst in calls.items():
        lst.sort()
        # the period: the smallest p for which the grid sizes repeat
        grids = [g for _, _, g in lst]
        period = next((p for p in range(1, len(grids) // 2 + 1) if all(grids[i] == grids[i % p] for i in range(len(grids)))
                       and len(grids) % p == 0), len(grids))
        print('%s: %d launches, period %d' % (key, len(lst), period))
        for k in range(period):
            d = [lst[i][1] for i in range(k, len(lst), period)]
            print('   #%2d grid %8d  median %8.1f us  (min %8.1f, %d samples)' % (k, grids[k], statistics.median(d) / 1e3,
                                                                                 min(d) / 1e3, len(d)))


if __name__ == '__main__':
    main()
