#!/usr/bin/env python3
"""Condense rocprofv3 output directories written by tools/profile_pmc.sh into a short text summary
(per-kernel average duration from the kernel trace; per-kernel mean counter values per dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict


def find(outdir, pattern):
    return sorted(glob.glob(os.path.join(outdir, "**", pattern), recursive=True))


def main():
    out = sys.argv[1]
    for f in find(os.path.join(out, "stats"), "*kernel_stats.csv"):
        print("== kernel stats:", os.path.relpath(f, out))
        for i, row in enumerate(csv.reader(open(f))):
            if i < 12:
                print("  ", ",".join(row)[:230])
    for f in find(os.path.join(out, "stats"), "*kernel_trace.csv"):
        durs = defaultdict(list)
        rd = csv.DictReader(open(f))
        for row in rd:
            try:
                durs[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
            except (KeyError, ValueError):
                pass
        print("== kernel trace:", os.path.relpath(f, out))
        for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1]))[:6]:
            print(f"   {k[:90]:90s} n={len(v):4d} avg={sum(v)/len(v)/1e3:10.1f} us min={min(v)/1e3:10.1f} us")
            # VGPR/LDS columns if present
    for p in ("pmc1", "pmc2", "pmc3", "pmc4"):
        for f in find(os.path.join(out, p), "*counter_collection.csv"):
            acc = defaultdict(lambda: defaultdict(list))
            meta = {}
            for row in csv.DictReader(open(f)):
                k = row.get("Kernel_Name", "?")
                acc[k][row.get("Counter_Name", "?")].append(float(row.get("Counter_Value", 0)))
                meta[k] = (row.get("VGPR_Count"), row.get("Accum_VGPR_Count"), row.get("LDS_Block_Size"), row.get("Grid_Size"))
            print("== counters:", os.path.relpath(f, out))
            for k, cs in acc.items():
                if "attn" not in k:
                    continue
                print(f"   {k[:100]}  vgpr/agpr/lds/grid={meta[k]}")
                for c, v in cs.items():
                    print(f"      {c:28s} mean/dispatch = {sum(v)/len(v):16.1f}   (n={len(v)})")


if __name__ == "__main__":
    main()
