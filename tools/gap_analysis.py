"""Scratch: per-kernel durations and inter-kernel gaps from a rocprofv3 kernel trace db."""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
rows = list(con.execute("select name, start, end, grid_x, grid_y from kernels where name like '%clo::%' order by start"))
rows = rows[len(rows)//2:]  # steady state
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for i, (n, s, e, gx, gy) in enumerate(rows):
    key = (n.split('(')[0].replace('void ', '')[:40], gx, gy)
    dur[key].append((e - s) / 1000)
    if i + 1 < len(rows):
        gap[key].append((rows[i + 1][1] - e) / 1000)
tot_d = tot_g = 0
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    d = sum(dur[k]) / len(dur[k]); g = sum(gap[k]) / max(len(gap[k]), 1)
    print(f"{k[0]:42s} grid=({k[1]},{k[2]}) dur {d:6.2f} us  gap-after {g:6.2f} us  n={len(dur[k])}")
    tot_d += d; tot_g += g
print(f"per matvec: kernels {tot_d:.1f} us + gaps {tot_g:.1f} us = {tot_d + tot_g:.1f} us")
