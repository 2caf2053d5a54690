#!/bin/bash
# per-kernel average durations of one bench workload via rocprofv3 --kernel-trace (run on the GPU box)
# usage: tools/time_kernels.sh <workload> [extra env assignments...]
W=${1:-fwdbwd_bf16_d128}; shift
export TMPDIR=/tmp
REPO=$(pwd)
OUT=/tmp/tk_$$
cd /tmp
env "$@" rocprofv3 --kernel-trace -d $OUT -o t -- python $REPO/bench.py --steps 5 --warmup 2 --workload $W --no-cpu-baseline > /dev/null 2>&1
python - <<PY
import sqlite3, glob
for f in glob.glob("$OUT/**/*.db", recursive=True):
    con = sqlite3.connect(f)
    for r in con.execute("select name, count(*), avg(end-start), min(end-start) from kernels where name like '%mfa%' group by name order by avg(end-start) desc"):
        print(f"{r[0][:80]:80s} n={r[1]} avg={r[2]/1e3:9.1f} us min={r[3]/1e3:9.1f} us")
PY
rm -rf $OUT
