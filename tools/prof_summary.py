"""Summarise a rocprofv3 results .db (kernel-trace --stats) into a small text table."""
import sqlite3, sys

def main(db, out, header):
    con = sqlite3.connect(db)
    rows = list(con.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
    with open(out, "w") as f:
        f.write(f"# {header}\n# calls   total_us    avg_us    pct  kernel\n")
        for n, c, t, a, p in rows:
            f.write(f"{c:7d} {t:10.1f} {a:9.2f} {p:6.2f}  {n[:120]}\n")

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], " ".join(sys.argv[3:]))
