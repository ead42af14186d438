#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 PMC counters from a rocpd sqlite db.  Usage: pmc_summary.py results.db [out.csv]"""
import re, sqlite3, sys
from collections import defaultdict

def main():
    db = sqlite3.connect(sys.argv[1])
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]
    ev, info, kd, ks = T("rocpd_pmc_event"), T("rocpd_info_pmc"), T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
    ecols = [r[1] for r in db.execute(f"pragma table_info({ev})")]
    scol = [r[1] for r in db.execute(f"pragma table_info({ks})")]
    name_col = "display_name" if "display_name" in scol else "kernel_name"
    q = (f"select s.{name_col}, p.name, sum(e.value), count(distinct d.id), sum(d.end-d.start)/count(distinct p.name) "
         f"from {ev} e join {info} p on e.pmc_id = p.id join {kd} d on e.event_id = d.event_id join {ks} s on d.kernel_id = s.id "
         f"group by s.{name_col}, p.name")
    agg = defaultdict(dict)
    calls, dur = {}, {}
    for kname, cname, val, n, _ in db.execute(q):
        kname = re.sub(r"\s+", " ", kname)
        agg[kname][cname] = val
        calls[kname] = n
    for kname, n, t in db.execute(f"select s.{name_col}, count(*), sum(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by s.{name_col}"):
        dur[re.sub(r"\s+", " ", kname)] = t
    counters = sorted({c for v in agg.values() for c in v})
    lines = ["kernel,calls,total_ms," + ",".join(counters)]
    for k in sorted(agg, key=lambda x: -dur.get(x, 0)):
        lines.append(f"\"{k[:90]}\",{calls[k]},{dur.get(k,0)/1e6:.3f}," + ",".join(f"{agg[k].get(c, 0):.6g}" for c in counters))
    out = "\n".join(lines)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out + "\n")
    print(out)

if __name__ == "__main__":
    main()
