#!/usr/bin/env python3
"""Print the kernel timeline of the last full step in a rocprofv3 rocpd database
(rocprofv3 --kernel-trace -d DIR -o NAME -> DIR/NAME_results.db)."""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = list(db.execute("select name,start,end,stream_id,grid_x from kernels order by start"))
idx = [i for i, r in enumerate(rows) if 'DriftKeys' in r[0]]
i0, i1 = idx[-2], idx[-1]
t0 = rows[i0][1]
for r in rows[i0:i1]:
    nm = re.sub(r'void psim::be::|psim::be::|psim_kernel<psim::|\(.*', '', r[0])
    print('%9.1f %8.1f s%d %-42s g=%d' % ((r[1] - t0) / 1e3, (r[2] - r[1]) / 1e3, r[3], nm[:42], r[4]))
print("launches in step:", i1 - i0)
