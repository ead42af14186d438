#!/usr/bin/env python3
"""Timeline of ONE render step from a rocprofv3 (rocpd sqlite) kernel trace: every dispatch between the last two
`sample_points_kernel` launches with its start offset, duration and queue — shows what actually overlaps on the side stream.
Usage: prof_timeline.py results.db [out.txt]"""
import re, sqlite3, sys


def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
    scol = [r[1] for r in db.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else "name")
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = f"select s.{name_col}, d.start, d.end" + (f", d.{qcol}" if qcol else ", 0") + f" from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"
    rows = list(db.execute(sel))
    marks = [i for i, r in enumerate(rows) if "sample_points_kernel" in r[0]]
    if len(marks) < 2:
        raise SystemExit("fewer than two steps in the trace")
    a, b = marks[-2], marks[-1]
    t0 = rows[a][1]
    out = [f"one step: {(rows[b][1] - t0) / 1e6:.3f} ms between two sample_points_kernel starts", "start_us   dur_us  queue  kernel"]
    for n, s, e, q in rows[a:b]:
        n = re.sub(r"\s+", " ", n)
        n = re.sub(r"\(anonymous namespace\)::", "", n)[:90]
        out.append(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:8.1f}  {q!s:>5}  {n}")
    txt = "\n".join(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(txt + "\n")
    print(txt)


if __name__ == "__main__":
    main()
