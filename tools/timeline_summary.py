#!/usr/bin/env python3
"""Per-kernel totals of the last full step in a rocprofv3 rocpd database, split by stream:
usage: timeline_summary.py DB"""
import collections
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end,stream_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'DriftKeys' in r[0]]
i0, i1 = idx[-2], idx[-1]
agg = collections.OrderedDict()
for name, s, e, st in rows[i0:i1]:
    nm = re.sub(r'void psim::be::|psim::be::|psim_kernel<psim::|\(.*', '', name)[:46]
    a = agg.setdefault((st, nm), [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
print("step span %.1f us, %d launches" % ((rows[i1][1] - rows[i0][1]) / 1e3, i1 - i0))
for (st, nm), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("  s%d %-46s x%5d %10.1f us" % (st, nm, c, t))
