#!/bin/bash
# same-box A/B of two builds of the library: new (in place), old (tools/ab/libpup_hip_old.so), alternating
cd "${GRAFT_REPO_ROOT:-/root/repo}"
cp coolpuppy_amd/libpup_hip.so /tmp/new.so
line() { timeout 300 python bench.py --steps 60 --warmup 5 --no-end-to-end --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$1', d['ms_per_step'], r.get('kernel_ms_per_launch'), r.get('prepass_ms_per_launch'))"; }
for rep in 1 2 3; do
  cp /tmp/new.so coolpuppy_amd/libpup_hip.so; line new
  cp tools/ab/libpup_hip_old.so coolpuppy_amd/libpup_hip.so; line old
done
cp /tmp/new.so coolpuppy_amd/libpup_hip.so
