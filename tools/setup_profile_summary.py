#!/usr/bin/env python3
"""Summarise the kernels of ONE per-frame setup (between the last KNN grid build's surroundings) in a rocprofv3 rocpd db."""
import sqlite3, re, sys
from collections import defaultdict
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
T = lambda p: [t for t in tabs if t.startswith(p)][0]
kd, ks = T("rocpd_kernel_dispatch"), T("rocpd_info_kernel_symbol")
rows = list(db.execute(f"select s.display_name,d.start,d.end from {kd} d join {ks} s on d.kernel_id=s.id order by d.start"))
idx = [i for i, r in enumerate(rows) if 'knn_bbox' in r[0]]
i0 = idx[-1]
j = i0
while j > 0 and 'composite_kernel' not in rows[j][0]: j -= 1
seg = rows[j + 1:]
k = [i for i, r in enumerate(seg) if 'point_fused_kernel' in r[0]][0]
seg = seg[:k]
agg = defaultdict(lambda: [0, 0.0])
for n, s, e in seg:
    n = re.sub(r'\(anonymous namespace\)::|void ', '', n)[:80]
    agg[n][0] += 1; agg[n][1] += (e - s) / 1e6
tot = sum(v[1] for v in agg.values())
print('per-frame setup: kernel time %.1f ms, span %.1f ms, %d launches' % (tot, (seg[-1][2] - seg[0][1]) / 1e6, len(seg)))
for n, v in sorted(agg.items(), key=lambda x: -x[1][1])[:20]:
    print('  %-80s x%-5d %.2f ms' % (n, v[0], v[1]))
