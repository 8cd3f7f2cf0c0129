#!/bin/bash
# A/B of library builds on the fitting step, same box, interleaved: bash tools/ab_fit.sh libA.so libB.so ...   (paths under ho-nerf_amd/)
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
  for L in "$@"; do
    export HONERF_LIB=$R/ho-nerf_amd/$L
    echo "$L: $(python3 $R/tools/fit_profile.py 60 2>/dev/null | tail -1)"
  done
done
