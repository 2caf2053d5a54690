#!/usr/bin/env python3
"""Condense the rocprofv3 (rocpd sqlite) outputs written by tools/profile_pmc.sh into a text summary:
per-kernel launch statistics from the kernel trace, mean PMC counter values per dispatch."""
import glob
import os
import sqlite3
import sys


def main():
    out = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else ""
    print(f"# rocprofv3 summary: {title}")
    print("# produced by tools/profile_pmc.sh (one --kernel-trace --stats pass + separate --pmc passes)")
    for db in sorted(glob.glob(os.path.join(out, "stats", "*.db"))):
        con = sqlite3.connect(db)
        print("\n## kernel trace (ns per dispatch)")
        q = ("select name, count(*), avg(end-start), min(end-start), max(end-start), max(vgpr_count), "
             "max(accum_vgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels "
             "where name like '%mfa%' group by name order by sum(end-start) desc")
        for r in con.execute(q):
            print(f"kernel={r[0]}\n  dispatches={r[1]} avg_ns={r[2]:.0f} min_ns={r[3]} max_ns={r[4]} "
                  f"vgpr={r[5]} agpr={r[6]} lds_bytes={r[7]} grid_x={r[8]} workgroup_x={r[9]}")
    for p in ("pmc1", "pmc2", "pmc3", "pmc4"):
        for db in sorted(glob.glob(os.path.join(out, p, "*.db"))):
            con = sqlite3.connect(db)
            print(f"\n## {p}: mean per dispatch")
            q = ("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                 "where kernel_name like '%mfa%' group by kernel_name, counter_name")
            last = None
            for r in con.execute(q):
                if r[0] != last:
                    print(f" kernel={r[0]}")
                    last = r[0]
                print(f"   {r[1]:28s} {r[2]:18.1f}  (n={r[3]})")


if __name__ == "__main__":
    main()
