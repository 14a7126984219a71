#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into the markdown tables kept
under profiles/.  Usage: python tools/rocpd_summary.py <results.db> [--by-grid] > profiles/<name>.md"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    by_grid = "--by-grid" in sys.argv
    c = sqlite3.connect(db)
    key = "name, grid_x, grid_y, grid_z" if by_grid else "name"
    # --split <substr>:<us>  reports launches of a kernel longer / shorter than <us> separately
    # (the flash kernel serves both the 48 832-key self-attention and the 512/257-key cross-attentions)
    name_expr = "name"
    for i, a in enumerate(sys.argv):
        if a == "--split":
            sub, us = sys.argv[i + 1].split(":")
            name_expr = (f"case when name like '%{sub}%' then name || (case when (end-start) > {float(us) * 1e3} "
                         f"then ' [>{us}us]' else ' [<={us}us]' end) else name end")
    key = key.replace("name", name_expr + " as kname", 1)
    grp = key.replace(name_expr + " as kname", "kname")
    rows = c.execute(
        f"select {key}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        f"max(vgpr_count), max(accum_vgpr_count), max(lds_size) from kernels group by {grp} order by sum(end-start) desc").fetchall()
    tot = sum(r[-7] for r in rows)
    print(f"rocprofv3 --kernel-trace --stats summary of `{db}`; total kernel time {tot / 1e6:.2f} ms\n")
    hdr = "| kernel |" + (" grid |" if by_grid else "") + " calls | total ms | avg us | min us | max us | % | vgpr | agpr | lds B |"
    print(hdr)
    print("|" + "---|" * hdr.count("|") )
    for r in rows:
        if by_grid:
            name, grid, rest = r[0], f"{r[1]}x{r[2]}x{r[3]}", r[4:]
        else:
            name, grid, rest = r[0], None, r[1:]
        n, s, a, mn, mx, vg, ag, lds = rest
        if s / tot < 0.0005:
            continue
        cells = [name[:60]] + ([grid] if by_grid else []) + [str(n), f"{s / 1e6:.2f}", f"{a / 1e3:.1f}", f"{mn / 1e3:.1f}",
                                                               f"{mx / 1e3:.1f}", f"{100 * s / tot:.1f}", str(vg), str(ag), str(lds)]
        print("| " + " | ".join(cells) + " |")


if __name__ == "__main__":
    main()
