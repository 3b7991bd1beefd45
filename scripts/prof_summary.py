#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (…_results.db) into the text summary committed under profiles/.

    python scripts/prof_summary.py gpurun_out/prof_r01/r01_results.db > profiles/r01_kernel_stats.txt
"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
print("# rocprofv3 --kernel-trace --stats summary (view top_kernels of %s)" % sys.argv[1].split("/")[-1])
print("%-60s %8s %16s %16s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
for name, calls, total, avg, pct in db.execute("select * from top_kernels"):
    print("%-60s %8d %16.3f %16.3f %8.3f" % (name[:60], calls, total, avg, pct))
try:
    rows = list(db.execute("select name, count(*), avg(value) from pmc_events group by name"))
    if rows:
        print("\n# PMC counters (per-dispatch average)")
        for r in rows:
            print("%-40s %8d %20.3f" % r)
except Exception:
    pass
