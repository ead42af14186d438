"""Summary of tools/knn_under_pf2.py's kernel trace: KNN launch durations by what ran beside them.  python tools/knn_under_pf2_summary.py results.db [out.txt]"""
import re, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
scol = [r[1] for r in db.execute(f"pragma table_info({ks})")]
name_col = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else "name")
rows = list(db.execute(f"select s.{name_col}, d.start, d.end from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"))
knn = [(s, e) for n, s, e in rows if "knn_wave_kernel" in n]
pf2 = [(s, e) for n, s, e in rows if "point_fused2_kernel" in n]
others = [(n, s, e) for n, s, e in rows if "knn_wave_kernel" not in n]
def overlap(a, b): return max(0, min(a[1], b[1]) - max(a[0], b[0]))
out = []
alone, under, mixed = [], [], []
for k in knn:
    dur = k[1] - k[0]
    ov_pf2 = sum(overlap(k, p) for p in pf2)
    ov_any = sum(overlap(k, (s, e)) for _, s, e in others)
    (under if ov_pf2 > 0.9 * dur else alone if ov_any < 0.05 * dur else mixed).append(dur / 1e3)
med = lambda v: sorted(v)[len(v) // 2] if v else float("nan")
out.append(f"half-batch KNN launches (262 144 queries): {len(knn)}")
out.append(f"  nothing else running:            n = {len(alone):3d}, median {med(alone):8.1f} us")
out.append(f"  entirely under point_fused2:     n = {len(under):3d}, median {med(under):8.1f} us")
out.append(f"  beside other kernels / partly:   n = {len(mixed):3d}, median {med(mixed):8.1f} us")
pf_alone = [(e - s) / 1e3 for s, e in pf2 if sum(overlap((s, e), k) for k in knn) < 0.05 * (e - s)]
pf_with = [(e - s) / 1e3 for s, e in pf2 if sum(overlap((s, e), k) for k in knn) > 0.9 * (e - s)]
out.append(f"point_fused2_kernel: alone n = {len(pf_alone)}, median {med(pf_alone):.1f} us; with searches beside it the whole time n = {len(pf_with)}, median {med(pf_with):.1f} us")
if under and pf_with and pf_alone and alone:
    gain = med(alone) - 0.0   # a half-batch search moved entirely under the matrix kernel leaves the critical path ...
    cost = (med(pf_with) - med(pf_alone)) * min(1.0, med(under) / med(pf_with))   # ... and the matrix kernel pays this while the search runs beside it
    out.append(f"a search that takes {med(alone):.0f} us alone takes {med(under):.0f} us under the matrix kernel, which runs {med(pf_with) - med(pf_alone):.0f} us longer per launch while searches run beside it")
txt = "\n".join(out)
if len(sys.argv) > 2: open(sys.argv[2], "w").write(txt + "\n")
print(txt)
