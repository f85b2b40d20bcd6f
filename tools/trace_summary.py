#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals for the LAST
physics step (from the last DriftKeys launch on), plus idle gaps between kernels."""
import csv, sys, glob, collections
f = sys.argv[1] if len(sys.argv) > 1 else glob.glob('gpurun_out/prof*/*/*kernel_trace.csv')[-1]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'DriftKeys' in r['Kernel_Name']]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
lo = starts[which]; hi = starts[which + 1] if which + 1 < 0 and which + 1 != 0 else len(rows)
if which + 1 < 0: hi = starts[which + 1]
seg = rows[lo:hi]
t0 = int(seg[0]['Start_Timestamp']); t1 = int(seg[-1]['End_Timestamp'])
agg = collections.OrderedDict()
busy = 0
prev_end = t0
gaps = []
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('void psim::be::', '').replace('psim::be::', '').replace('psim_kernel<psim::', '<')[:60]
    a = agg.setdefault(name, [0, 0])
    a[0] += 1; a[1] += e - s
    busy += e - s
    if s - prev_end > 15000: gaps.append((s - prev_end, name))
    prev_end = max(prev_end, e)
print("step span %.3f ms, kernels busy %.3f ms, %d launches" % ((t1 - t0) / 1e6, busy / 1e6, len(seg)))
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-62s x%4d %9.1f us" % (k, c, t / 1e3))
print("gaps > 15us before kernel: total %.1f us in %d gaps" % (sum(g for g, _ in gaps) / 1e3, len(gaps)))
for g, nm in sorted(gaps, reverse=True)[:12]:
    print("   %8.1f us before %s" % (g / 1e3, nm))
