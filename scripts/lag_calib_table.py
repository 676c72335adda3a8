#!/usr/bin/env python3
"""Table from scripts/lag_calib.py's output: per kept solve, excess PCG iterations against the movement measures."""
import re
import sys

base = name = start = None
for line in open(sys.argv[1]):
    m = re.match(r"== (.*): refresh always \[(.*)\]", line)
    if m:
        name, base, start = m.group(1), [int(x) for x in m.group(2).split(",")], None
        continue
    m = re.match(r"-- (.*): kept from iteration (\d+) on", line)
    if m:
        start = int(m.group(2))
        continue
    m = re.search(r"iteration= (\d+)\s+pcg= (\d+).*diag moved (\S+) \((\d+) rows > 1/4\), weights: (\d+) edges > 1/5, (\d+) > 1/2, sum (\S+?)( coarse|$)", line)
    if m and start is not None and m.group(8).strip():
        it = int(m.group(1))
        print(f"{name:24s} kept since {start:2d} it {it:2d} pcg {int(m.group(2)):4d} fresh {base[it]:3d} excess {int(m.group(2)) - base[it]:+4d}  "
              f"diag {float(m.group(3)):.2e} rows {m.group(4):>4s}  w>1/5 {m.group(5):>5s} w>1/2 {m.group(6):>5s} sumw {float(m.group(7)):.2e}")
