#!/bin/bash
# instrumented build (make EXTRA=-DXRIT_RELAY_TIMING): where the W-wave walker's step goes
cd "$(dirname "$0")/.." || exit 1
for W in ${WAVES:-4 8}; do
  echo "=== XRIT_RELAY_WAVES=$W"
  XRIT_RELAY_WAVES=$W XRIT_TRACE=1 timeout 300 python scripts/relay_burst.py --log2 28 --bursts 2 --exact ${EXACT:-0} --prof 2>&1 | grep "relay team\|relay pass\|clock_relay\|burst" | tail -${TAILN:-14}
done
