#!/usr/bin/env python3
"""HBM traffic per render step from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected SEPARATELY: they do not fit one
pass on gfx950) over the same bench command -> profiles/<tag>_hbm_traffic.json with the fingerprint of the kernel sources.

  python tools/hbm_traffic.py fetch.db write.db steps_in_run out.json
FETCH_SIZE / WRITE_SIZE are reported in KB by rocprofv3; FETCH_SIZE is doubled per the gfx950 note of MI355X_MICROARCH.md."""
import json, os, sqlite3, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def total(dbpath, counter):
    db = sqlite3.connect(dbpath)
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
    T = lambda p: [t for t in tabs if t.startswith(p)][0]
    ev, info = T("rocpd_pmc_event"), T("rocpd_info_pmc")
    return db.execute(f"select sum(e.value) from {ev} e join {info} p on e.pmc_id = p.id where p.name = ?", (counter,)).fetchone()[0]


def main():
    fdb, wdb, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    prec = sys.argv[5] if len(sys.argv) > 5 else "f16mx"
    import bench
    f_kb, w_kb = total(fdb, "FETCH_SIZE"), total(wdb, "WRITE_SIZE")
    d = {"config": f"c2 {prec}", "precision": prec, "render_steps_in_run": steps,
         "fetch_GB_per_step_raw": f_kb * 1024 / steps / 1e9,
         "fetch_GB_per_step_x2_gfx950_correction": 2 * f_kb * 1024 / steps / 1e9,
         "write_GB_per_step": w_kb * 1024 / steps / 1e9,
         "sources_sha": bench.sources_sha(),
         "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over `bench.py --steps 3 --warmup 1 --no-cpu-baseline --also ''` "
                 "(7 render steps incl. the dominant-kernel pass), summed over all kernels / steps; FETCH_SIZE on gfx950 counts 128-B requests "
                 "as 64 B for wide coalesced streams (MI355X_MICROARCH.md HBM section), so the x2 figure is the upper estimate; Infinity-Cache hits are included"}
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(d, indent=1))


if __name__ == "__main__":
    main()
