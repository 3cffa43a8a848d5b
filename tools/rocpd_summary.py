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
    base = m.group(1) if m else name.replace("(anonymous namespace)::", "").split("(")[0][:200]
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
    busy_marker = None
    if "--busy" in args:
        k = args.index("--busy"); busy_marker = args[k + 1]; args = args[:k] + args[k + 2:]
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


    if busy_marker:
        # GPU-busy analysis of an optimisation loop (bench.py --workload c4_loop / vertex_loop): one step = the interval between two launches of `busy_marker`
        # (a kernel that runs once per step).  Per step: wall span, the UNION of all kernel intervals (kernels of two streams overlap: their sum is not a time),
        # idle = span - union, and the kernel time by family (the library's har:: kernels, the update kernels, torch's, runtime copies / fills).
        db = sqlite3.connect(args[0])
        rows = db.execute("select name, start, end from kernels order by start").fetchall()
        marks = [i for i, r in enumerate(rows) if busy_marker in r[0]]
        steps = []
        for a, b in zip(marks[:-1], marks[1:]):
            seg = rows[a:b]
            span = rows[b][1] - rows[a][1]
            union = 0; cur_s, cur_e = seg[0][1], seg[0][2]
            for _, st, en in seg[1:]:
                if st > cur_e:
                    union += cur_e - cur_s; cur_s, cur_e = st, en
                else:
                    cur_e = max(cur_e, en)
            union += cur_e - cur_s
            fam = {}
            for n, st, en in seg:
                sh = short(n)
                key = ("update" if any(x in sh for x in ("k_set_positions", "k_vertex_normals", "k_shading_triangles", "k_refit")) else
                       "library" if "har" in n else "torch" if "at::" in n else "runtime")
                fam[key] = fam.get(key, 0) + (en - st)
            steps.append((span, union, fam, len(seg)))
        # drop the longest spans (the boundaries between the bench's phases hold host-side setup)
        if steps:
            med = sorted(s[0] for s in steps)[len(steps) // 2]
            keep = [s for s in steps if s[0] < 1.5 * med]
            n = len(keep)
            print("-- GPU busy per step (%d steps between launches of %s; %d phase boundaries dropped)" % (n, busy_marker, len(steps) - n))
            span = sum(s[0] for s in keep) / n / 1e6; union = sum(s[1] for s in keep) / n / 1e6
            print("   span %.3f ms  union of kernel intervals %.3f ms  idle %.3f ms (%.1f %%)  launches %.0f" % (span, union, span - union, 100 * (span - union) / span, sum(s[3] for s in keep) / n))
            fams = sorted(set(k for s in keep for k in s[2]))
            print("   kernel time by family (sums; two streams overlap): " + ", ".join("%s %.3f ms" % (k, sum(s[2].get(k, 0) for s in keep) / n / 1e6) for k in fams))
            summary["_busy"] = {"steps": n, "span_ms": span, "union_ms": union, "idle_ms": span - union, "by_family_ms": {k: sum(s[2].get(k, 0) for s in keep) / n / 1e6 for k in fams}}
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
