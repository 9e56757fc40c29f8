#!/bin/bash
# walkers per sweep (4 pairs, host-driven loop) for two certificate margins
cd "$(dirname "$0")/.." || exit 1
for m in 1e-5 2e-6; do
  echo "=== LH_CERT_REL=$m"
  LH_CERT_REL=$m LH_WALK_LOG=1 LH_PROBE_PAIRS=4 LH_PROBE_SOLVER=1 timeout 60 python tools/probe_iter_times.py 2>&1 | grep "lh walks" | tail -80 | awk '{w[$4]=w[$4]" "$8} END {for (s in w) print "slot", s, ":", w[s]}'
done
