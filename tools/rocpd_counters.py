#!/usr/bin/env python
"""Per-kernel mean of the PMC counters in a rocprofv3 (ROCm 7.2, rocpd sqlite) --pmc run.
usage: python tools/rocpd_counters.py <results.db> [name-substring] [--by-grid]   -> one line per (kernel[, grid], counter)"""
import sqlite3
import sys


def main():
    c = sqlite3.connect(sys.argv[1])
    args = [a for a in sys.argv[2:] if a != "--by-grid"]
    by_grid = "--by-grid" in sys.argv[2:]
    sub = args[0] if args else ""
    cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
    cnt_col = "counter_name" if "counter_name" in cols else "counter"
    val_col = "value" if "value" in cols else "counter_value"
    disp = "dispatch_id" if "dispatch_id" in cols else None
    if name_col is None:
        print("columns:", cols)
        return
    grid_col = next((c_ for c_ in ("grid_size", "grid_size_x", "grid_x") if c_ in cols), None)
    if by_grid and not grid_col:
        print("(no grid column among", cols, ": per-kernel means)")
    if by_grid and grid_col:             # one line per (kernel, grid, counter): the shapes of one kernel apart
        q = (f"select {name_col}, {grid_col}, {cnt_col}, avg(v), count(*) from (select {name_col}, {grid_col}, {cnt_col}, {disp}, sum({val_col}) as v "
             f"from counters_collection where {name_col} like ? group by {name_col}, {grid_col}, {cnt_col}, {disp}) group by {name_col}, {grid_col}, {cnt_col}")
        for name, gr, cn, v, n in c.execute(q, (f"%{sub}%",)):
            print(f"{name[:60]:60s} grid {gr:<10} {cn:14s} mean/launch {v:.6g}  launches {n}")
        return
    # a counter is reported per dimension instance (XCD / SE ...): sum them per dispatch, then average the dispatches
    q = (f"select {name_col}, {cnt_col}, avg(v), count(*) from (select {name_col}, {cnt_col}, {disp}, sum({val_col}) as v "
         f"from counters_collection where {name_col} like ? group by {name_col}, {cnt_col}, {disp}) group by {name_col}, {cnt_col}")
    for name, cn, v, n in c.execute(q, (f"%{sub}%",)):
        print(f"{name[:70]:70s} {cn:28s} mean/launch {v:.6g}  launches {n}")


if __name__ == "__main__":
    main()
