#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) result: per-kernel launch count / total / average
duration, and per-kernel sums of collected PMC counters.  Usage: rocpd_summary.py file.db [...]"""
import re
import sqlite3
import sys


def short(name):
    m = re.search(r"har\d*(k_[a-z_]+)", name)
    # demangled names: the whole template-id (two flavours of k_shade differ in their LAST arguments -- cut at 60 characters they shared one key, and the second
    # overwrote the first's counters in the JSON summary: bench.py then found half the dispatches it expected and quoted no traffic figure)
    base = m.group(1) if m else name.split("(")[0][:200]
    t = re.search(r"ILi(\d+)(?:EL[ij](\d+))?", name)
    if m and t:
        base += "<" + ",".join(x for x in t.groups() if x) + ">"
    return base


def main():
    import json
    args = sys.argv[1:]
    json_out = None
    if "--json" in args:
        k = args.index("--json"); json_out = args[k + 1]; args = args[:k] + args[k + 2:]
    launches_out = None
    if "--launches" in args:
        k = args.index("--launches"); launches_out = args[k + 1]; args = args[:k] + args[k + 2:]
    summary = {}
    for path in args:
        db = sqlite3.connect(path)
        cur = db.cursor()
        print("== %s" % path)
        cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
        rows = cur.execute("select name, count(*), sum(end - start), avg(end - start), min(end-start), max(end-start) from kernels group by name order by 3 desc").fetchall()
        tot = sum(r[2] for r in rows) or 1
        print("%-44s %8s %12s %12s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "%"))
        for n, c, s, a, mn, mx in rows:
            print("%-44s %8d %12.3f %12.2f %10.2f %10.2f %6.1f" % (short(n), c, s / 1e6, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))
            summary.setdefault(short(n), {}).update({"calls": c, "total_ms": s / 1e6, "avg_us": a / 1e3})
        try:
            pm = cur.execute("select k.name, p.counter_name, sum(p.value), count(*) from counters_collection p join kernels k on 1=0").fetchall()
        except Exception:
            pm = []
        try:
            ccols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
            if ccols:
                namecol = "kernel_name" if "kernel_name" in ccols else ("name" if "name" in ccols else None)
                q = "select %s, counter_name, sum(value), count(*) from counters_collection group by 1, 2 order by 1, 2" % namecol
                rows = cur.execute(q).fetchall()
                if rows:
                    print("-- PMC sums per kernel (sum over dispatches, dispatch count)")
                    for n, cn, v, c in rows:
                        print("%-44s %-24s %20.0f %8d" % (short(n), cn, v, c))
                        summary.setdefault(short(n), {}).setdefault("counters", {})[cn] = {"sum": v, "dispatches": c}
        except Exception as e:
            print("(no counters: %s)" % e)


    if launches_out:
        # per-launch table of the LAST frame (dispatch order): one line per kernel launch from the last k_raygen on
        db = sqlite3.connect(args[0])
        rows = db.execute("select name, start, end from kernels order by start").fetchall()
        last = max((i for i, r in enumerate(rows) if "k_raygen" in r[0]), default=0)
        with open(launches_out, "w") as f:
            f.write("# kernel launches of the last frame in dispatch order: start offset (us), duration (us), gap to the previous launch's end (us)\n")
            t0, prev_end = rows[last][1], rows[last][1]
            for n, st, en in rows[last:]:
                f.write("%-40s %10.1f %10.1f %8.1f\n" % (short(n), (st - t0) / 1e3, (en - st) / 1e3, (st - prev_end) / 1e3))
                prev_end = en
            f.write("# frame: %.1f us from the first launch's start to the last launch's end\n" % ((prev_end - t0) / 1e3))
    if json_out:
        with open(json_out, "w") as f:
            json.dump(summary, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
