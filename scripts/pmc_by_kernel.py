#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counters from a rocpd database.

    python scripts/pmc_by_kernel.py <..._results.db> [kernel-substring]
Prints, for every kernel (optionally filtered), dispatch count, total duration and the counter totals per dispatch.
Works off the rocpd views when present, else off the raw tables (schema differs a little between ROCm builds)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
def find(prefix):
    c = [n for n in names if n == prefix] or [n for n in names if n.startswith(prefix)]
    return c[0] if c else None
if "--schema" in sys.argv:
    for n in names:
        print(n, [c[1] for c in db.execute("pragma table_info('%s')" % n)])
    sys.exit(0)
v = find("counters_collection")
if v:
    cols = [c[1] for c in db.execute("pragma table_info('%s')" % v)]
    kcol = "kernel_name" if "kernel_name" in cols else "name"
    rows = db.execute("select %s, counter_name, count(*), sum(value) from %s group by 1, 2" % (kcol, v)).fetchall()
else:
    ev, info, kd, ks = find("rocpd_pmc_event"), find("rocpd_info_pmc"), find("rocpd_kernel_dispatch"), find("rocpd_info_kernel_symbol")
    rows = db.execute("select s.kernel_name, i.name, count(*), sum(e.value) from %s e join %s i on e.pmc_id = i.id "
                      "join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by 1, 2" % (ev, info, kd, ks)).fetchall()
agg = {}
for k, c, n, s in rows:
    if flt and flt not in k: continue
    agg.setdefault(k, {})[c] = (n, s)
for k, cs in agg.items():
    print(k[:70])
    for c, (n, s) in sorted(cs.items()):
        print("    %-28s dispatches(x dims) %6d   total %18.0f   per dispatch-row %16.1f" % (c, n, s, s / n))
