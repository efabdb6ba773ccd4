"""List the launches of ONE inner step (single task) from a rocprofv3 --kernel-trace CSV: start offset, duration, gap to the previous
launch, grid, kernel.  usage: step_timeline.py <kernel_trace.csv> [step_index_from_end=2]
The step boundary is the clip_sgd launch (one per inner step)."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows.sort(key=lambda r: int(r['Start_Timestamp']))
ends = [i for i, r in enumerate(rows) if 'clip_sgd' in r['Kernel_Name']]
lo, hi = ends[-back - 1] + 1, ends[-back] + 1
# all_shadows follows clip_sgd: count it with the step it belongs to
while hi < len(rows) and ('all_shadows' in rows[hi]['Kernel_Name']): hi += 1
while 'all_shadows' in rows[lo]['Kernel_Name']: lo += 1
t0 = int(rows[lo]['Start_Timestamp']); prev_end = t0; busy = 0
for r in rows[lo:hi]:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    g = f"{int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}x{int(r['Grid_Size_Y'])//int(r['Workgroup_Size_Y'])}x{int(r['Grid_Size_Z'])//int(r['Workgroup_Size_Z'])}"
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f}us gap {(s - prev_end) / 1e3:6.1f}  wg {g:12s} {name[:90]}")
    busy += e - s; prev_end = e
print(f"step: {(prev_end - t0) / 1e3:.1f} us wall, {busy / 1e3:.1f} us in kernels, {hi - lo} launches")
