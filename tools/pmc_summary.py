"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (two separate runs) into a small JSON +
text table: HBM-side bytes per launch for every libclo kernel.

Correction per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE reports
exactly half of the bytes of a wide coalesced streaming read, so fetch bytes = 2 * FETCH_SIZE KiB;
WRITE_SIZE is taken as is (KiB).
"""
import json, sqlite3, sys

def collect(db, counter):
    con = sqlite3.connect(db)
    q = ("select kernel_name, grid_size_x, grid_size_y, count(*), avg(value) from counters_collection "
         "where counter_name=? and kernel_name like '%clo::%' group by kernel_name, grid_size_x, grid_size_y")
    return {(n.split('(')[0].replace('void ', ''), gx, gy): (c, v) for n, gx, gy, c, v in con.execute(q, (counter,))}

def main(fetch_db, write_db, out_json, out_txt, header):
    f, w = collect(fetch_db, "FETCH_SIZE"), collect(write_db, "WRITE_SIZE")
    rows = []
    for key in sorted(set(f) | set(w), key=lambda k: -(f.get(k, (0, 0))[1] * 2 + w.get(k, (0, 0))[1])):
        fk, wk = f.get(key, (0, 0.0)), w.get(key, (0, 0.0))
        rows.append({"kernel": key[0], "grid": [key[1], key[2]], "launches": fk[0] or wk[0],
                     "fetch_bytes": 2 * fk[1] * 1024, "write_bytes": wk[1] * 1024,
                     "hbm_bytes": 2 * fk[1] * 1024 + wk[1] * 1024})
    import glob, hashlib, os
    h = hashlib.sha256()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in sorted(glob.glob(os.path.join(root, "curvlinops_amd", "csrc", "*.hip"))):
        h.update(open(f, "rb").read())
    json.dump({"note": header, "csrc_sha16": h.hexdigest()[:16], "kernels": rows}, open(out_json, "w"), indent=1)
    with open(out_txt, "w") as t:
        t.write(f"# {header}\n# fetch = 2 x FETCH_SIZE KiB (gfx950 correction), write = WRITE_SIZE KiB; bytes per launch\n")
        t.write("# launches   fetch_MB   write_MB   total_MB  kernel [grid]\n")
        for r in rows:
            t.write(f"{r['launches']:8d} {r['fetch_bytes']/1e6:10.2f} {r['write_bytes']/1e6:10.2f} {r['hbm_bytes']/1e6:10.2f}  {r['kernel']} {r['grid']}\n")

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], " ".join(sys.argv[5:]))
