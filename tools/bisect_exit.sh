#!/bin/bash
# which test file makes the interpreter abort at exit? (GPUTEST rc 134)  Usage: tools/bisect_exit.sh [out-prefix]
out=${1:-gpurun_out/bisect}
for f in tests/test_baseline_configs_gpu.py tests/test_callbacks.py tests/test_coverage_gpu.py tests/test_dist_gloo.py tests/test_edge_cases.py tests/test_golden_gpu.py tests/test_kernel_parity.py tests/test_properties_gpu.py; do
  python3 -m pytest $f -x -q -m gpu -p no:cacheprovider > ${out}_$(basename $f .py).log 2>&1
  echo "$f rc=$?" | tee -a ${out}_summary.txt
done
