#!/usr/bin/env python3
"""Row-owner partition of a config, host only (sgo_plan_rows: no GPU): boundary rows per world size, i.e. what one
exchange of the multi-GPU row-owner mode moves (DESIGN.md section 6's model table).
Usage: python scripts/boundary_model.py [C4] [2 4 8]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sparse_gslam_amd import capi, synth  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "C4"
    worlds = [int(a) for a in sys.argv[2:]] or [2, 4, 8]
    g = synth.config(name)
    for G in worlds:
        plan = capi.plan_rows(g.poses, g.fixed, g.ei, g.ej, G)
        n = plan["n"]
        row_of = np.full(g.V, -1, dtype=np.int64)
        row_of[plan["row_vertex"]] = np.arange(n)
        rb = plan["rank_row_begin"]
        ri, rj = row_of[g.ei], row_of[g.ej]
        ok = (ri >= 0) & (rj >= 0)
        ri, rj = ri[ok], rj[ok]
        qi, qj = np.searchsorted(rb, ri, side="right") - 1, np.searchsorted(rb, rj, side="right") - 1
        cross = qi != qj
        isb = np.zeros(n, dtype=bool)
        isb[ri[cross]] = True
        isb[rj[cross]] = True
        per = [int(isb[rb[q]:rb[q + 1]].sum()) for q in range(G)]
        rows = [int(rb[q + 1] - rb[q]) for q in range(G)]
        bmax = max(per)
        print(f"{name} G={G}: tiles {plan['tile_row_begin'].size - 1}, rows per rank {min(rows)}..{max(rows)}, boundary rows total {sum(per)} "
              f"({100.0 * sum(per) / n:.2f} % of n), largest rank {bmax}; cross-rank edges {int(cross.sum())} of {ri.size}; "
              f"packet (4 scalars + 3 x {bmax} doubles) {8 * (4 + 3 * bmax) / 1024:.1f} KiB per rank, "
              f"all-gather delivers {8 * (4 + 3 * bmax) * G / 1024:.1f} KiB; all-reduce mode moves {24 * n / 1024:.0f} KiB per product",
              flush=True)


if __name__ == "__main__":
    main()
