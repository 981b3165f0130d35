#!/bin/bash
# A/B timing of variant libraries (tools/variant.sh) against the shipped one: alternates them, two rounds.
#   bash tools/ab_variants.sh "<variant names, '-' = shipped>" <probe.py> [probe args]
R=$(cd "$(dirname "$0")/.." && pwd); V=${MFG_VARIANT_DIR:-$R/discrete_mean_field_game_amd/csrc/variants}
names=$1; shift
for round in 1 2; do
  for n in $names; do
    if [ "$n" = "-" ]; then unset MFG_HIP_LIB; else export MFG_HIP_LIB=$V/lib$n.so; fi
    echo "== $n (round $round)"
    python "$@" 2>&1 | grep -v "^$"
  done
done
