#!/bin/bash
# Runs on the GPU box: scripts/streamed_timeline.sh with each library variant under xritdemod_amd/lib/ab/ in turn; the relay and
# front-end kernels of one steady-state burst per variant.  Output: gpurun_out/ab_timeline.txt
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp xritdemod_amd/lib/libxritdemod_amd.so /tmp/lib_orig.so
: > gpurun_out/ab_timeline.txt
for v in xritdemod_amd/lib/ab/*.so; do
  cp $v xritdemod_amd/lib/libxritdemod_amd.so
  scripts/streamed_timeline.sh 0 > /dev/null 2>&1
  echo "== $(basename $v)" >> gpurun_out/ab_timeline.txt
  grep -E "one burst|clock_relay|fir_decim|copyBuffer|scan_reduce_kernel<UnwrapF>" gpurun_out/streamed_timeline.txt >> gpurun_out/ab_timeline.txt
done
cp /tmp/lib_orig.so xritdemod_amd/lib/libxritdemod_amd.so
cat gpurun_out/ab_timeline.txt
