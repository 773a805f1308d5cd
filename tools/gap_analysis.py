"""Scratch: per-kernel durations and inter-kernel gaps from a rocprofv3 kernel trace db.
usage: gap_analysis.py results.db [kernel-substring-launched-once-per-matvec]"""
import sqlite3, sys, collections
con = sqlite3.connect(sys.argv[1])
once = sys.argv[2] if len(sys.argv) > 2 else None
rows = list(con.execute("select name, start, end, grid_x, grid_y from kernels where name like '%clo::%' order by start"))
rows = rows[len(rows)//2:]  # steady state
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for i, (n, s, e, gx, gy) in enumerate(rows):
    key = (n.split('(')[0].replace('void ', '')[:44], gx, gy)
    dur[key].append((e - s) / 1000)
    if i + 1 < len(rows):
        gap[key].append((rows[i + 1][1] - e) / 1000)
nmv = sum(len(v) for k, v in dur.items() if once and once in k[0]) or min(len(v) for v in dur.values())
tot_d = tot_g = 0
for k in sorted(dur, key=lambda k: -sum(dur[k])):
    d = sum(dur[k]) / len(dur[k]); g = sum(gap[k]) / max(len(gap[k]), 1)
    per = len(dur[k]) / nmv
    print(f"{k[0]:46s} grid=({k[1]},{k[2]}) dur {d:7.2f} us x {per:4.1f}/mv = {d*per:7.1f}  gap-after {g:5.2f}")
    tot_d += d * per; tot_g += g * per
print(f"per matvec ({nmv} matvecs): kernels {tot_d:.1f} us + gaps {tot_g:.1f} us = {tot_d + tot_g:.1f} us")
