#!/usr/bin/env python3
"""Print the kernel timeline of the last ray chunk in a rocprofv3 rocpd db."""
import sqlite3, re, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
kd, ks = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
rows = list(db.execute(f"select s.display_name, d.start, d.end, d.grid_size_x, d.grid_size_y from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
idx = [i for i, r in enumerate(rows) if 'sample_points' in r[0]]
i0 = idx[-1]
tot = 0
for r in rows[i0:i0+70]:
    n = re.sub(r'\(anonymous namespace\)::|void ', '', r[0])[:34]
    print(f"{n:34s} {(r[2]-r[1])/1e3:9.1f} us  grid=({r[3]//256},{r[4]})")
    tot += r[2]-r[1]
    if 'gemm' in n and rows[i0:i0+70].index(r) > 30 and False: pass
    if 'composite' in n:
        nxt = rows[rows.index(r)+1] if rows.index(r)+1 < len(rows) else None
        if nxt and 'gemm' in nxt[0]:
            print(f"{'gemm (feat, per ray)':34s} {(nxt[2]-nxt[1])/1e3:9.1f} us"); tot += nxt[2]-nxt[1]
        break
print('chunk total', round(tot/1e3, 1), 'us')
