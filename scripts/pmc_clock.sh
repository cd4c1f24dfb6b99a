#!/bin/bash
# Runs on the GPU box: instruction mix of the clock-recovery kernels over one C2 burst.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sqc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_BRANCH --output-format csv -d $OUT/p1 -o c2 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-profile "$@" > $OUT/p1.log 2>&1
python - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/p1/c2_counter_collection.csv")))
per = collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    k = r["Kernel_Name"]
    if "clock_pass_kernel" in k or "clock_output_kernel" in k or "costas_pass" in k:
        per[(k.split("(")[0][-44:], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
for (k, d), v in sorted(per.items(), key=lambda kv: int(kv[0][1])):
    w = max(v.get("SQ_WAVES", 1), 1)
    print(k, d, "waves %d" % w, {c: round(x / w, 1) for c, x in v.items() if c != "SQ_WAVES"})
PY
