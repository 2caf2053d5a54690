#!/bin/bash
# Vendor GEMM under the counters (run on the GPU box via gpurun): matrix-pipe busy fraction and effective clock of hipBLASLt's bf16
# kernel on N(0,1) and on zeros, next to the un-profiled TF/s -- the calibration of "what a pure matrix-instruction body sustains on
# this box" that the attention kernel's roofline fraction is read against.
#   usage: tools/vendor_calib.sh <outdir>
set -u
OUT=${1:-gpurun_out/vendor_calib}
export TMPDIR=/tmp
REPO=$(pwd)
mkdir -p "$OUT"
python tools/vendor_calib.py > "$OUT/unprofiled.txt" 2>&1
for FILL in normal zero; do
  cd /tmp
  rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS GRBM_GUI_ACTIVE \
      -d "$REPO/$OUT/pmc_$FILL" -o pmc -- python "$REPO/tools/vendor_calib.py" --fills $FILL --rounds 2 --iters 20 > "$REPO/$OUT/pmc_$FILL.log" 2>&1
  cd "$REPO"
done
python - "$OUT" <<'PY' > "$OUT/vendor_calib.txt" 2>&1
import glob, os, sqlite3, sys
out = sys.argv[1]
print("# vendor GEMM calibration (tools/vendor_calib.sh): hipBLASLt bf16 n = 8192 A*B^T")
print(open(os.path.join(out, "unprofiled.txt")).read())
for fill in ("normal", "zero"):
    for db in sorted(glob.glob(os.path.join(out, "pmc_" + fill, "*.db"))):
        con = sqlite3.connect(db)
        # the dominant kernel = the GEMM
        rows = list(con.execute("select name, count(*), avg(end-start) from kernels group by name order by sum(end-start) desc limit 1"))
        if not rows:
            continue
        name, n, avg_ns = rows[0]
        print(f"## fill = {fill}: kernel {name[:100]}  dispatches {n}  avg {avg_ns / 1e3:.1f} us (profiled)")
        vals = {}
        for r in con.execute("select counter_name, avg(value) from counters_collection where kernel_name = ? group by counter_name", (name,)):
            vals[r[0]] = r[1]
            print(f"   {r[0]:28s} {r[1]:18.1f}")
        if "GRBM_GUI_ACTIVE" in vals and "SQ_VALU_MFMA_BUSY_CYCLES" in vals:
            clk = vals["GRBM_GUI_ACTIVE"] / 8.0
            print(f"   -> effective clock {clk / avg_ns:.3f} GHz, matrix pipe busy {vals['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024.0 / clk:.3f} of the shader clocks")
PY
rm -rf "$OUT/pmc_normal" "$OUT/pmc_zero"
cat "$OUT/vendor_calib.txt"
