#!/bin/bash
# counter passes (two rocprofv3 --pmc runs each) of ONE headline-shaped launch on the dense stream, the geometry stream without the mask and the
# causal stream of attn_fwd16_p4p: what profiles/r06_counters_dense_vs_geometry_vs_causal.txt holds (run inside one gpurun call)
cd "$(dirname "$0")/.." || exit 1
OUT=gpurun_out/r06_geom; mkdir -p $OUT; export TMPDIR=/tmp; REPO=$(pwd)
for M in dense geom causal; do
  cd /tmp
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE -d "$REPO/$OUT/pmc1_$M" -o pmc -- python "$REPO/tools/counters_one_launch.py" $M > "$REPO/$OUT/$M.log" 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_BRANCH -d "$REPO/$OUT/pmc2_$M" -o pmc -- python "$REPO/tools/counters_one_launch.py" $M >> "$REPO/$OUT/$M.log" 2>&1
  cd "$REPO"
done
python - "$OUT" <<'PY' > "$OUT/summary.txt" 2>&1
import glob, os, sqlite3, sys
out = sys.argv[1]
for m in ("dense", "geom", "causal"):
    for p in ("pmc1", "pmc2"):
        for db in sorted(glob.glob(os.path.join(out, f"{p}_{m}", "*.db"))):
            con = sqlite3.connect(db)
            rows = list(con.execute("select name, count(*), avg(end-start) from kernels where name like '%p4p%' group by name order by sum(end-start) desc limit 1"))
            if not rows: continue
            name, n, avg = rows[0]
            print(f"## {m} {p}: {name[:70]} n={n} avg {avg/1e3:.1f} us")
            for r in con.execute("select counter_name, avg(value) from counters_collection where kernel_name = ? group by counter_name", (name,)):
                print(f"   {r[0]:28s} {r[1]:18.1f}")
PY
rm -rf $OUT/pmc1_* $OUT/pmc2_*
cat $OUT/summary.txt
