#!/bin/bash
# Dev tool (GPU box): K1q with both tiles of a pair in every wave (tools/ab/libpup_hip_dual.so, -DPUP_K1Q_DUAL=1) against the build in place.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp coolpuppy_amd/libpup_hip.so /tmp/base.so
for rep in 1 2 3; do
  for which in base dual; do
    if [ $which = base ]; then cp /tmp/base.so coolpuppy_amd/libpup_hip.so; else cp tools/ab/libpup_hip_dual.so coolpuppy_amd/libpup_hip.so; fi
    echo "$which $(timeout 200 python tools/k1_probe.py --variants 0 --reps 9 2>&1 | grep k1_ms | cut -c1-120)"
  done
done
cp tools/ab/libpup_hip_dual.so coolpuppy_amd/libpup_hip.so
timeout 900 python -m pytest tests/test_properties_gpu.py tests/test_kernel_parity.py tests/test_golden_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python tools/k1_probe.py --variants 67108864 --reps 3 2>&1 | grep phases | cut -c1-400
cp /tmp/base.so coolpuppy_amd/libpup_hip.so
