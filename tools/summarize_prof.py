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


def write_traffic_json(out, workload):
    """profiles/traffic.json: HBM bytes per launch of the dominant kernel = 2 x FETCH_SIZE + WRITE_SIZE (KiB; FETCH_SIZE doubled
    as MI355X_MICROARCH.md prescribes for wide coalesced reads on gfx950), tied to the SHA-256 of the library that was
    profiled -- bench.py reports the number only while that library is the one it loads."""
    import hashlib
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    from metal_flash_attention_amd import _abi
    vals = {}
    for p, name in (("pmc3", "FETCH_SIZE"), ("pmc4", "WRITE_SIZE")):
        for db in sorted(glob.glob(os.path.join(out, p, "*.db"))):
            con = sqlite3.connect(db)
            q = ("select kernel_name, avg(value), count(*) from counters_collection where counter_name = ? and "
                 "kernel_name like '%mfa%' group by kernel_name order by sum(value) desc limit 1")
            for r in con.execute(q, (name,)):
                vals[name] = (r[0], r[1])
    if len(vals) != 2:
        print("traffic.json not written: counters missing", vals)
        return
    kernel = vals["FETCH_SIZE"][0]
    # matrix-pipe busy cycles and GPU-active cycles of the same kernel (pmc1): bench.py derives `mfma_busy` from them
    extra = {}
    for db in sorted(glob.glob(os.path.join(out, "pmc1", "*.db"))):
        con = sqlite3.connect(db)
        for r in con.execute("select counter_name, avg(value) from counters_collection where kernel_name = ? group by counter_name", (kernel,)):
            extra[r[0]] = r[1]
    bytes_per_launch = (2.0 * vals["FETCH_SIZE"][1] + vals["WRITE_SIZE"][1]) * 1024.0
    from metal_flash_attention_amd import AttentionKernel  # noqa: F401  (loads the library named below)
    digest = hashlib.sha256(open(_abi.library_path(), "rb").read()).hexdigest()
    variant = os.environ.get("MFA_PROFILED_VARIANT", "")
    path = os.path.join(root, "profiles", "traffic.json")
    try:
        table = json.load(open(path))
    except Exception:  # noqa: BLE001
        table = {"entries": []}
    table["entries"] = [e for e in table["entries"] if not (e.get("workload") == workload and e.get("variant") == variant)]
    table["entries"].append({"workload": workload, "variant": variant, "kernel": kernel, "lib_sha256": digest,
                             "bytes_per_launch": bytes_per_launch, "fetch_size_kib": vals["FETCH_SIZE"][1],
                             "write_size_kib": vals["WRITE_SIZE"][1],
                             "mfma_busy_cycles": extra.get("SQ_VALU_MFMA_BUSY_CYCLES"), "gui_active": extra.get("GRBM_GUI_ACTIVE"),
                             "insts_mfma": extra.get("SQ_INSTS_MFMA"),
                             "source": os.path.join(out, "summary.txt") + " (copied to profiles/)"})
    json.dump(table, open(path, "w"), indent=1)
    print("wrote", path, "-", bytes_per_launch, "bytes/launch for", variant)


if __name__ == "__main__":
    main()
    if len(sys.argv) > 3:
        write_traffic_json(sys.argv[1], sys.argv[3])
