#!/usr/bin/env python3
"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel calls / total / mean / share.  Usage: prof_summary.py results.db [out.csv]"""
import re, sqlite3, sys

def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
    scol = [r[1] for r in db.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else "name")
    q = f"select s.{name_col}, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.{name_col} order by 3 desc"
    rows = list(db.execute(q))
    tot = sum(r[2] for r in rows) or 1
    lines = ["kernel,calls,total_ms,mean_us,min_us,max_us,share_pct"]
    for n, c, t, mn, mx in rows:
        n = re.sub(r"\s+", " ", n)
        lines.append(f"\"{n}\",{c},{t/1e6:.3f},{t/c/1e3:.2f},{mn/1e3:.2f},{mx/1e3:.2f},{100*t/tot:.2f}")
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    print(out)

if __name__ == "__main__":
    main()
