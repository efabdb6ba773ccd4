#!/usr/bin/env python3
"""Where is the GPU idle?  Reads a rocprofv3 kernel trace (…_kernel_trace.csv), merges the kernel intervals of all streams and
reports the time during which NO kernel was running: total, histogram by gap length and, for the long gaps, which kernel ended
before and which started after them (i.e. which host-side phase the GPU was waiting for).

    rocprofv3 --kernel-trace -d gpurun_out/e2e_trace -o t --output-format csv -- python3 tools/bench_pretrain.py --configs host:4 --meta-steps 20
    python tools/gap_analysis.py gpurun_out/e2e_trace [--skip-first 0.3] [--min-gap-us 30]
"""
import argparse
import collections
import csv
import glob
import json
import os
import sys


def short(n):
    tag = n[n.rfind(" ["):] if n.endswith("]") and " [q" in n else ""
    if tag: n = n[:n.rfind(" [")]
    return _short(n) + tag


def _short(n):
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    return n.split("(")[0][:48]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("dir")
    ap.add_argument("--skip-first", type=float, default=0.3, help="fraction of the trace (by time) dropped at the start (warm-up, load_data)")
    ap.add_argument("--min-gap-us", type=float, default=30.0)
    ap.add_argument("--out", default=None)
    ap.add_argument("--window", default=None, help="also print the kernel timeline around the N-th launch of this kernel (name substring)")
    ap.add_argument("--window-index", type=int, default=10)
    ap.add_argument("--window-ms", type=float, default=2.5)
    args = ap.parse_args()
    files = sorted(glob.glob(os.path.join(args.dir, "**", "*kernel_trace.csv"), recursive=True), key=os.path.getmtime)
    if not files:
        sys.exit("no kernel trace under " + args.dir)
    rows = []
    with open(files[-1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"] + (" [q%s s%s t%s]" % (r.get("Queue_Id"), r.get("Stream_Id"), r.get("Thread_Id")) if os.environ.get("GAP_QUEUES") else "")))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    cut = t0 + int((t1 - t0) * args.skip_first)
    rows = [r for r in rows if r[0] >= cut]
    span = max(r[1] for r in rows) - rows[0][0]
    busy_end, last_name = rows[0][1], rows[0][2]
    idle = 0
    hist = collections.Counter()
    pairs = collections.defaultdict(lambda: [0, 0])
    kernel_time = 0
    for s, e, n in rows:
        kernel_time += e - s
        if s > busy_end:
            g = s - busy_end
            idle += g
            b = "<10us" if g < 10e3 else "10-30us" if g < 30e3 else "30-100us" if g < 100e3 else "100-300us" if g < 300e3 else "0.3-1ms" if g < 1e6 else ">1ms"
            hist[b] += g
            if g >= args.min_gap_us * 1e3:
                p = pairs[(short(last_name), short(n))]
                p[0] += 1; p[1] += g
        if e > busy_end:
            busy_end, last_name = e, n
    res = {"trace": files[-1], "span_ms": span / 1e6, "idle_ms": idle / 1e6, "idle_frac": idle / span, "avg_kernels_in_flight": kernel_time / (span - idle),
           "idle_ms_by_gap_length": {k: v / 1e6 for k, v in hist.items()},
           "long_gaps": [{"after": a, "before": b, "count": c, "total_ms": t / 1e6, "avg_us": t / c / 1e3}
                         for (a, b), (c, t) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:14]]}
    if args.window:
        hits = [r for r in rows if args.window in r[2]]
        if hits:
            c = hits[min(args.window_index, len(hits) - 1)][0]
            w = args.window_ms * 1e6
            tl, prev_end = [], None
            for s_, e_, n_ in rows:
                if c - w <= s_ <= c + w:
                    tl.append("%9.1f us  +%7.1f  gap %7.1f  %s" % ((s_ - c) / 1e3, (e_ - s_) / 1e3, (s_ - prev_end) / 1e3 if prev_end else 0.0, short(n_)))
                prev_end = e_ if prev_end is None else max(prev_end, e_)
            res["timeline"] = tl
            # every gap > 60 us between this launch and the next one of the same kernel (one period of the loop), and the
            # GPU's busy fraction / kernels in flight per 0.5 ms bucket of that period
            nxt = hits[min(args.window_index, len(hits) - 1) + 1][0] if min(args.window_index, len(hits) - 1) + 1 < len(hits) else c + int(w)
            per, be, ln = [], None, None
            nb = int((nxt - c) / 5e5) + 1
            busy = [0.0] * nb; infl = [0.0] * nb
            for s_, e_, n_ in rows:
                if e_ < c or s_ > nxt:
                    if s_ <= nxt: be = e_ if be is None else max(be, e_)
                    continue
                if be is not None and s_ - be > 60e3 and s_ >= c:
                    per.append("at %8.1f us: idle %7.1f us, then %s (after %s)" % ((be - c) / 1e3, (s_ - be) / 1e3, short(n_), ln))
                if be is None or e_ > be:
                    lo = max(s_, be if be is not None else s_, c)
                    for b in range(nb):
                        a0, a1 = c + b * 5e5, c + (b + 1) * 5e5
                        ov = min(e_, a1) - max(lo, a0)
                        if ov > 0: busy[b] += ov
                    be, ln = e_, short(n_)
                for b in range(nb):
                    a0, a1 = c + b * 5e5, c + (b + 1) * 5e5
                    ov = min(e_, a1) - max(s_, a0)
                    if ov > 0: infl[b] += ov
            res["period_ms"] = (nxt - c) / 1e6
            res["period_gaps"] = per
            res["busy_frac_per_0.5ms"] = [round(x / 5e5, 2) for x in busy]
            res["kernels_in_flight_per_0.5ms"] = [round(x / 5e5, 1) for x in infl]
    txt = json.dumps(res, indent=1)
    print(txt)
    if args.out:
        open(args.out, "w").write(txt + "\n")


if __name__ == "__main__":
    main()
