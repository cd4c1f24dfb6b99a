#!/bin/bash
# the W-wave relay walker: bit-for-bit against the serial wave, then timing on C2 bursts
cd "$(dirname "$0")/.." || exit 1
for W in ${WAVES:-2 4}; do
  echo "=== XRIT_RELAY_WAVES=$W (2^24 samples, serial comparison)"
  XRIT_RELAY_WAVES=$W timeout 300 python scripts/relay_burst.py --log2 24 --bursts 2 --serial --exact 1 0 2>&1 | grep "burst\|rror" | tail -12
done
for W in 1 ${WAVES:-2 4}; do
  echo "=== XRIT_RELAY_WAVES=$W (C2 bursts)"
  XRIT_RELAY_WAVES=$W XRIT_TRACE=1 timeout 300 python scripts/relay_burst.py --log2 28 --bursts 3 --exact 1 0 --prof 2>&1 | grep "relay team\|relay pass [0-2]:\|clock_relay \|burst\|rror" | tail -24
done
