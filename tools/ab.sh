# usage: bash tools/ab.sh <variant-name> <probe.py> [args...]   -- alternates the shipped library and a variant, twice
V=$1; shift
for rep in 1 2; do
  echo "== shipped"; python "$@" 2>&1 | grep -v "^$"
  echo "== $V"; MFG_HIP_LIB=discrete_mean_field_game_amd/csrc/variants/lib$V.so python "$@" 2>&1 | grep -v "^$"
done
