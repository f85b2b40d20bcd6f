#!/usr/bin/env python3
"""Achieved HBM rate of the streaming kernels of one step.

usage: hbm_table.py TRACE_DB [FETCH_DB [WRITE_DB]] [--n PARTICLES]

TRACE_DB: rocprofv3 --kernel-trace --output-format rocpd run of `bench.py --no-overlap`
(every stage on one stream, so durations are not stretched by the walk).  FETCH_DB /
WRITE_DB: separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` runs of the same command.
Algorithmic bytes per particle are listed in ROWS below (state arrays read and written by
each kernel, FP64 positions as 16-byte pairs).  Counter values are
printed raw (KiB as rocprofv3 reports them -> bytes); on gfx950 FETCH_SIZE reads half the
bytes of a wide coalesced stream and is uncalibrated for narrower or scattered accesses."""
import re
import sqlite3
import sys

args = list(sys.argv[1:])
n = 1 << 22
if "--n" in args:
    at = args.index("--n")
    n = int(args[at + 1])
    del args[at:at + 2]
PEAK = 8000.0   # GB/s, MI355X HBM3E specification

# kernel-name substring -> (label, algorithmic bytes per particle)
# (bytes that every launch moves whatever the data: stores that only happen for some
# particles -- wall hits -- are left out, so the figure is a lower bound on the traffic)
ROWS = [
    ("DriftKeys", "DriftKeys (drift + Morton key + cell id + packed record)", 128),
    ("sort_pass_kernel<unsigned long, 16>", "Morton sort pass (u64 key + u32 index)", 24),
    ("sort_pass_kernel<unsigned int, 16>", "cell sort pass (u32 key + u32 index)", 16),
    ("FixRuns", "FixRuns (in place: keys read, only runs rewritten)", 8),
    ("GatherTree", "GatherTree (random 32 B gather)", 60),
    ("GatherCell", "GatherCell (random 32 B gather)", 72),
    ("HilbertCellRuns", "HilbertCellRuns", 8),
    ("HilbertOrder", "HilbertOrder", 12),
    ("TreeLinks", "TreeLinks", 13),
    ("tree_emit_tile_kernel", "tree_emit_tile_kernel (elements + in-tile centres of mass)", 97),
    ("Kick", "Kick", 64),
    ("WallsFlag", "WallsFlag (+ conduction delta, cooling)", 61),
]
ACHIEVABLE = 6300.0   # GB/s: above this the inputs were (partly) served by the 256 MiB Infinity Cache


def last_step(db, value_sql):
    rows = list(db.execute(value_sql))
    idx = [i for i, r in enumerate(rows) if "DriftKeys" in r[0]]
    return rows[idx[-2]:idx[-1]]


trace = sqlite3.connect(args[0])
kern = last_step(trace, "select name, start, end, grid_x from kernels order by start")


def counter_of(path, name):
    if not path:
        return {}
    db = sqlite3.connect(path)
    q = ("select kernel_name, dispatch_id, sum(value) from counters_collection where counter_name = '%s' "
         "group by 2 order by 2" % name)
    rows = list(db.execute(q))
    idx = [i for i, r in enumerate(rows) if "DriftKeys" in r[0]]
    out = {}
    for k, _, v in rows[idx[-2]:idx[-1]]:
        out.setdefault(k, []).append(v)
    return out


fetch = counter_of(args[1] if len(args) > 1 else None, "FETCH_SIZE")
write = counter_of(args[2] if len(args) > 2 else None, "WRITE_SIZE")

print("%-60s %5s %9s %9s %7s %6s %10s %10s" % ("kernel", "calls", "us/call", "alg MB", "GB/s", "frac", "FETCH MB", "WRITE MB"))
for sub, label, bpp in ROWS:
    full = (n + 4095) // 4096 * 256 if "sort_pass" in sub else n      # launches over all n particles only
    d = [(r[2] - r[1]) / 1e3 for r in kern if sub in r[0] and full <= r[3] < full + 512]
    if not d:
        continue
    us = sum(d) / len(d)
    mb = bpp * n / 1e6
    f = [v for k, vs in fetch.items() if sub in k for v in vs]
    w = [v for k, vs in write.items() if sub in k for v in vs]
    fs = "%10.1f" % (max(f) * 1024 / 1e6) if f else "%10s" % "-"
    ws = "%10.1f" % (max(w) * 1024 / 1e6) if w else "%10s" % "-"
    if bpp:
        gbs = mb / 1e3 / (us * 1e-6)
        if gbs > ACHIEVABLE:
            # not an HBM rate: inputs written by the kernels just before are still on the die
            print("%-60s %5d %9.1f %9.1f %7.0f %6s %s %s  (inputs resident in the Infinity Cache: no HBM fraction)" % (
                label, len(d), us, mb, gbs, "-", fs, ws))
        else:
            print("%-60s %5d %9.1f %9.1f %7.0f %6.3f %s %s" % (label, len(d), us, mb, gbs, gbs / PEAK, fs, ws))
    else:
        print("%-60s %5d %9.1f %9s %7s %6s %s %s" % (label, len(d), us, "-", "-", "-", fs, ws))
