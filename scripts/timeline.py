#!/usr/bin/env python3
"""Kernel timeline of a rocprofv3 --kernel-trace database: every dispatch longer than a threshold, in start order.

    python scripts/timeline.py results.db [min_ms=20]
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 20.0
views = [r[0] for r in db.execute("select name from sqlite_master where type in ('view','table')")]
view = "kernels" if "kernels" in views else None
if view is None:
    print("no `kernels` view; tables:", views)
    sys.exit(1)
cols = [r[1] for r in db.execute("pragma table_info(%s)" % view)]
rows = list(db.execute("select name, start, end, queue_id, stream_id from %s order by start" % view)) if "stream_id" in cols else \
       [r + (0, 0) for r in db.execute("select name, start, end from %s order by start" % view)]
t0 = rows[0][1]
for name, st, en, qid, sid in rows:
    if (en - st) / 1e6 >= min_ms:
        print("%9.1f ms  +%9.1f ms  q%-3s s%-3s %s" % ((st - t0) / 1e6, (en - st) / 1e6, qid, sid, name[:70]))
