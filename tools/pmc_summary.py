#!/usr/bin/env python3
"""Sum the counters of a rocprofv3 --pmc run (rocpd database) per kernel dispatch.
usage: pmc_summary.py DB [kernel-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
want = sys.argv[2] if len(sys.argv) > 2 else "gravity_walk_wave_kernel"
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
if "counters_collection" in tabs:
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    q = "select dispatch_id, kernel_name, counter_name, sum(value) from counters_collection group by 1,2,3 order by 1,3"
    last = None
    for did, k, c, v in db.execute(q):
        if want not in k:
            continue
        if did != last:
            print("dispatch", did, k[:60])
            last = did
        print("   %-28s %.0f" % (c, v))
else:
    print("tables:", tabs)
