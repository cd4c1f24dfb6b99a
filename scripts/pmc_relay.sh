#!/bin/bash
# Runs on the GPU box: SQ counters of the relay kernels (one-wave walker against the teams), one counter pass each
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sq_relay
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for W in ${WAVES:-1 2 4}; do
  XRIT_RELAY_WAVES=$W rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/a$W -o r -- python $R/scripts/relay_burst.py --log2 28 --bursts 2 --exact 0 > $OUT/a$W.log 2>&1
  XRIT_RELAY_WAVES=$W rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD --output-format csv -d $OUT/b$W -o r -- python $R/scripts/relay_burst.py --log2 28 --bursts 2 --exact 0 > $OUT/b$W.log 2>&1
done
python - $OUT <<'PY'
import collections, csv, glob, sys
for f in sorted(glob.glob(sys.argv[1] + "/*/**/*counter_collection.csv", recursive=True)):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("xrit::", "")[:40]
        if "clock_relay" in k and "init" not in k and "finalize" not in k:
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print("==", f.split("sq_relay/")[1].split("/")[0])
    for k, v in per.items():
        print(k, {c: round(sum(x) / len(x)) for c, x in sorted(v.items())}, "launches", max(len(x) for x in v.values()))
PY
rm -rf $OUT/a? $OUT/b?
