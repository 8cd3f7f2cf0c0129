#!/bin/bash
# A/B of library builds on the fitting step with per-kernel times (kernel trace): bash tools/ab_fit_kernels.sh libA.so libB.so
R=${GRAFT_REPO_ROOT:-$(pwd)}
export TMPDIR=/tmp
cd /tmp
for L in "$@"; do
  export HONERF_LIB=$R/ho-nerf_amd/$L
  rm -rf /tmp/abk_$L
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk_$L -- python3 $R/tools/fit_profile.py 20 > /dev/null 2>&1
  T=$(find /tmp/abk_$L -name "*kernel_trace.csv" | head -1)
  echo "== $L"
  python3 $R/tools/trace_gaps.py $T 12 4 | grep -E "steps|k_field2_(hand|obj)<[34]>"
done
