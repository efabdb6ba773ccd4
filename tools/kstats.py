"""Summarise a rocprofv3 --kernel-trace --stats CSV: ms per inner step per kernel.  usage: kstats.py <kernel_stats.csv> <steps> [top]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 40
out = []
for r in rows:
    n = int(r['Calls']); t = float(r['TotalDurationNs'])
    if n < steps:
        continue
    out.append((t / steps / 1e6, n / steps, t / n / 1e3, r['Name'][:100]))
out.sort(reverse=True)
print(f"total kernel time {sum(o[0] for o in out):.3f} ms/step over {sum(o[1] for o in out):.0f} launches/step")
cum = 0
for ms, c, avg, name in out[:top]:
    cum += ms
    print(f"{ms:6.3f} {c:5.1f} {avg:7.1f}us  cum {cum:5.2f}  {name}")
