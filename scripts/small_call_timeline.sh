#!/bin/bash
# Runs on the GPU box: every launch of one steady-state call at a reference-sized chunk (default 2^19 samples, d=5).
LOG2=${1:-19}; D=${2:-5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/tl_small
cat > /tmp/one_small.py <<PY
import sys, os, time
sys.path.insert(0, "$R")
import torch
import xritdemod_amd as xa
from xritdemod_amd import _capi
n = 1 << $LOG2; D = $D; fs = 1.25e6 * D
sp = _capi.synth_params(fs_in=fs)
buf = torch.empty((12, n, 2), dtype=torch.float32, device="cuda:0")
st = torch.cuda.current_stream().cuda_stream
for b in range(12):
    _capi.synth_generate_device(sp, b * n, n, buf[b].data_ptr(), device=0, stream=st)
dem = xa.Demodulator(xa.Demodulator.config("lrit", fs, D))
soft = torch.empty((n,), dtype=torch.float32, device="cuda:0")
for b in range(12):
    dem.process_device(buf[b].data_ptr(), n, soft.data_ptr(), n, stream=st)
torch.cuda.synchronize()
PY
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/tl_small -o t -- python /tmp/one_small.py > /dev/null 2>&1
F=$(find $R/gpurun_out/tl_small -name 't_kernel_trace.csv' | head -1)
python $R/scripts/timeline_print.py "$F" "${3:-fir_decim_kernel<3}"
