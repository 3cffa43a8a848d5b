#!/usr/bin/env python3
"""Per-launch durations (us) of the traversal kernels of the LAST frame in a rocprofv3 rocpd database, in launch order."""
import sqlite3, sys
con = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in con.execute("select name from sqlite_master where type='table'")]
def tab(prefix): return [t for t in tabs if t.startswith(prefix)][0]
kd, ks = tab("rocpd_kernel_dispatch"), tab("rocpd_info_kernel_symbol")
scols = {r[1] for r in con.execute("pragma table_info(%s)" % ks)}
name_col = "display_name" if "display_name" in scols else "kernel_name"
rows = con.execute("select s.%s, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (name_col, kd, ks)).fetchall()
seq = [(n, (e - s) / 1e3) for n, s, e in rows if "k_trace_closest" in n or "k_resolve" in n or "k_shade" in n]
per = int(sys.argv[2]) if len(sys.argv) > 2 else 24
last = seq[-per:]
for kind in ("k_trace_closest", "k_shade", "k_resolve"):
    print(kind, " ".join("%.0f" % d for n, d in last if kind in n))
