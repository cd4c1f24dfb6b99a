#!/bin/bash
# Runs on the GPU box: SQ counter passes over one C2 burst (where do the waves of each kernel spend their cycles?)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/p1 -o c2 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-profile "$@" > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_ACTIVE_INST_SCA --output-format csv -d $OUT/p2 -o c2 -- python $R/bench.py --steps 1 --warmup 1 --no-cpu --no-profile "$@" > $OUT/p2.log 2>&1
ls -R $OUT | head
