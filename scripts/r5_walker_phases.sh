#!/bin/bash
# Runs on the GPU box with a library built with EXTRA="-DXRIT_EXPERIMENTS -DXRIT_RELAY_TIMING" at xritdemod_amd/lib/ab/timing.so:
# shader-clock cycles per 64-symbol step the overlap walkers spend in each phase, one burst at a time against the streamed pipeline.
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
cp xritdemod_amd/lib/libxritdemod_amd.so /tmp/lib_orig.so
cp xritdemod_amd/lib/ab/timing.so xritdemod_amd/lib/libxritdemod_amd.so
for mode in "--no-prefetch" ""; do
  echo "== bench.py $mode $@"
  XRIT_WALKER_PHASES=1 python bench.py --steps 30 --warmup 4 --no-cpu --no-serial-floor --no-exact --no-profile $mode "$@" 2>&1 | grep -E "overlap walkers|ms_per_step" | sed -e 's/.*"ms_per_step": \([0-9.]*\).*/ms_per_step \1/' | tail -8
done
cp /tmp/lib_orig.so xritdemod_amd/lib/libxritdemod_amd.so
