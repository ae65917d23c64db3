#!/bin/bash
# dev: same-box A/B of the wide kernel (new build in place, old = tools/ab/libpup_hip_old.so) over a few window widths
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp coolpuppy_amd/libpup_hip.so /tmp/new.so
for p in 16 20 25 40 50; do
  for which in new old; do
    if [ $which = new ]; then cp /tmp/new.so coolpuppy_amd/libpup_hip.so; else cp tools/ab/libpup_hip_old.so coolpuppy_amd/libpup_hip.so; fi
    echo "pad $p $which $(timeout 200 python tools/k1_probe.py --variants 0 --reps 5 --pad $p 2>&1 | tail -1 | cut -c1-120)"
  done
done
cp /tmp/new.so coolpuppy_amd/libpup_hip.so
