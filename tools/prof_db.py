#!/usr/bin/env python3
"""Per-kernel count / average duration (ms) from a rocprofv3 results .db (the default output format of this ROCm)."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start)/1e6, min(d.end-d.start)/1e6 from {kd} d join {ks} s "
     f"on d.kernel_id=s.id group by s.kernel_name order by 3 desc")
for r in c.execute(q):
    print(f"{r[0][:100]:100s} n={r[1]:4d} avg={r[2]:.3f} ms min={r[3]:.3f} ms")
