#!/usr/bin/env python3
"""Per-queue busy time and idle gaps from a rocprofv3 kernel trace (``--kernel-trace --output-format csv``).

    python tools/summarize_gaps.py <..._kernel_trace.csv> [--skip-first N]

For every hardware queue: number of kernels, sum of kernel durations, span from the first start to the last end, and the
distribution of the gaps between one kernel's end and the next one's start on that queue -- the part of a launch-bound
step that no kernel optimisation touches."""
import argparse
import csv
import json
import statistics
from collections import defaultdict


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('trace')
    ap.add_argument('--skip-first', type=float, default=0.5, help='fraction of the trace (by time) to drop as warm-up')
    args = ap.parse_args()
    rows = list(csv.DictReader(open(args.trace)))
    start_key = next(k for k in rows[0] if k.lower() in ('start_timestamp', 'start'))
    end_key = next(k for k in rows[0] if k.lower() in ('end_timestamp', 'end'))
    queue_key = next((k for k in rows[0] if k.lower() in ('queue_id', 'queue')), None)
    t0 = min(int(r[start_key]) for r in rows)
    t1 = max(int(r[end_key]) for r in rows)
    cut = t0 + args.skip_first * (t1 - t0)
    queues = defaultdict(list)
    for r in rows:
        if int(r[start_key]) >= cut:
            queues[r[queue_key] if queue_key else '0'].append((int(r[start_key]), int(r[end_key]), r.get('Kernel_Name', r.get('kernel_name', ''))))
    out = {}
    for q, ks in queues.items():
        ks.sort()
        busy = sum(e - s for s, e, _ in ks)
        gaps = [max(ks[i + 1][0] - ks[i][1], 0) for i in range(len(ks) - 1)]
        span = ks[-1][1] - ks[0][0]
        by_name = defaultdict(lambda: [0, 0])
        for s, e, n in ks:
            by_name[n.split('(')[0][:60]][0] += 1
            by_name[n.split('(')[0][:60]][1] += e - s
        out[q] = {'kernels': len(ks), 'busy_ms': round(busy / 1e6, 3), 'span_ms': round(span / 1e6, 3),
                  'busy_frac': round(busy / max(span, 1), 3), 'avg_kernel_us': round(busy / len(ks) / 1e3, 2),
                  'gap_us_median': round(statistics.median(gaps) / 1e3, 2) if gaps else None,
                  'gap_us_mean': round(statistics.mean(gaps) / 1e3, 2) if gaps else None,
                  'top': sorted(((n, c, round(t / 1e6, 3)) for n, (c, t) in by_name.items()), key=lambda r: -r[2])[:12]}
    # union over queues: time during which at least one kernel was running
    allk = sorted((s, e) for ks in queues.values() for s, e, _ in ks)
    union, cur_s, cur_e = 0, allk[0][0], allk[0][1]
    for s, e in allk[1:]:
        if s > cur_e:
            union += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    union += cur_e - cur_s
    total_span = allk[-1][1] - allk[0][0] if len(allk) > 1 else 1
    print(json.dumps({'device_busy_ms': round(union / 1e6, 3), 'span_ms': round(total_span / 1e6, 3),
                      'device_busy_frac': round(union / total_span, 3)}))
    for q, d in out.items():
        top = d.pop('top')
        print(q, json.dumps(d))
        for n, c, t in top:
            print(f'    {t:9.3f} ms {c:6d} x {1e3 * t / c:8.2f} us  {n}')


if __name__ == '__main__':
    main()
