"""Summary of tools/knn_under_pf2.py's kernel trace: how much search work gets done beside each kernel of the step, and what it costs that kernel.
python tools/knn_under_pf2_summary.py results.db [out.txt]

CAVEAT (found the hard way, round 6): the "progress" column spreads a search's work evenly over its duration.  A search that starts beside a gather kernel (fast) and
ends under the neural-point kernel (5 x slower: one of its workgroups fits a CU there) has most of its progress credited to the wrong window, so the column OVERSTATES
what runs beside the matrix kernel by ~4 x.  The in-situ measurement is profiles/r6_knn_parts_*_timeline.txt (the search released in parts under the neural-point
launches: 262 144 queries take 1.97 ms there and cost the kernel +0.34 ms).  The slow-down columns (kernel alone | with searches beside it) are sound."""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scol = [r[1] for r in db.execute(f"pragma table_info({ks})")]
name_col = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else "name")
rows = list(db.execute(f"select s.{name_col}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
short = lambda n: re.sub(r"\(anonymous namespace\)::", "", re.sub(r"\s+", " ", n)).split("(")[0].replace("void ", "")[:48]
knn = [(s, e) for n, s, e in rows if "knn_wave_kernel" in n]
others = [(short(n), s, e) for n, s, e in rows if "knn_wave_kernel" not in n]
def overlap(a, b): return max(0, min(a[1], b[1]) - max(a[0], b[0]))
med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
# searches with nothing else on the device = the search alone
alone = [(e - s) / 1e3 for s, e in knn if sum(overlap((s, e), (a, b)) for _, a, b in others) < 0.02 * (e - s)]
t_alone = med(alone)
out = [f"half-batch exact KNN (262 144 queries), nothing else running: {t_alone:.0f} us (n = {len(alone)})",
       "per kernel of the render step: its duration alone | with searches beside it the whole time | search progress per launch of it, in search-alone microseconds",
       "(progress = sum over the searches of overlap / that search's duration x the search's alone time: what the side stream got done while this kernel ran)"]
names = []
for n, _, _ in others:
    if n not in names: names.append(n)
tot_gain = tot_cost = 0.0
for nm in names:
    iv = [(s, e) for n, s, e in others if n == nm]
    if med([(e - s) / 1e3 for s, e in iv]) < 30: continue
    d_alone = [(e - s) / 1e3 for s, e in iv if sum(overlap((s, e), k) for k in knn) < 0.02 * (e - s)]
    busy = [(s, e) for s, e in iv if sum(overlap((s, e), k) for k in knn) > 0.95 * (e - s)]
    d_with = [(e - s) / 1e3 for s, e in busy]
    prog = [sum(overlap((s, e), k) / (k[1] - k[0]) for k in knn) * t_alone for s, e in busy]
    if not d_alone or not d_with: continue
    out.append(f"  {nm:48s} {med(d_alone):8.0f} | {med(d_with):8.0f} (+{med(d_with) - med(d_alone):5.0f}) | {med(prog):7.0f}")
txt = "\n".join(out)
if len(sys.argv) > 2: open(sys.argv[2], "w").write(txt + "\n")
print(txt)
