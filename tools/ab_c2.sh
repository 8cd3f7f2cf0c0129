#!/bin/bash
# A/B of library builds on the C2 frame (the dominant kernel timed inside the timed steps) and on the fitting step:
#   bash tools/ab_c2.sh libhonerf.so libhonerf_ab_x.so ...     (libraries under ho-nerf_amd/, built with make BUILD=... OUT=... CXXFLAGS_EXTRA=...)
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for rep in 1 2; do
for L in "$@"; do
  export HONERF_LIB=$R/ho-nerf_amd/$L
  python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-culled --no-fitting --no-training --no-c1 --no-f16 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline())
r = d['roofline']
print('%-28s C2 kernel %.2f ms (in the timed steps), step %.2f ms, %.2f M ray-samples/s, sustained MFMA probe %.0f TFLOP/s' % ('$L', r['kernel_ms'], d['ms_per_step'], d['value'] / 1e6, r.get('mfma_sustained_tflops', 0)))"
  python3 tools/fit_profile.py 60 2>&1 | tail -1 | sed "s/^/$L  fitting step /"
done
done
