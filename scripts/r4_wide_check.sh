#!/bin/bash
# first runs of the W-wave relay walker: bit-for-bit against the serial wave, then timing on a C2 burst
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
for W in 4 8 2; do
  echo "=== XRIT_RELAY_WAVES=$W (2^24 samples, serial comparison)"
  XRIT_RELAY_WAVES=$W XRIT_TRACE=1 timeout 300 python scripts/relay_burst.py --log2 24 --bursts 2 --serial --exact 1 3 2>&1 | grep -v "clock pass\|walker " | tail -40
done
for W in 1 4 8; do
  echo "=== XRIT_RELAY_WAVES=$W (C2 burst)"
  XRIT_RELAY_WAVES=$W XRIT_TRACE=1 timeout 300 python scripts/relay_burst.py --log2 28 --bursts 3 --exact 1 0 --prof 2>&1 | grep -v "clock pass\|walker " | tail -60
done
