#!/bin/bash
# Dev tool (GPU box): the sparse trans kernel (configs[4]) over its compile-time knobs — queue slots in flight (PUP_K1S_U) and filter
# words in flight (PUP_K1S_NA) — and chunk sizes; parity tests first.  Usage: bash tools/ab/k1s_sweep.sh "U NA" ["U NA" ...]
OUT=gpurun_out/k1s_sweep; mkdir -p $OUT; export TRANS_CACHE=/tmp/trans_cache.npz
timeout 400 python -m pytest tests/test_kernel_parity.py tests/test_baseline_configs_gpu.py tests/test_float_values_gpu.py -q -m gpu -k "trans or sparse or config4 or float" 2>&1 | tail -3 > $OUT/tests.txt
timeout 400 python tools/probe_trans.py ${CHUNKS:-0,120,250} > $OUT/probe_default.txt 2>&1
for cfg in "$@"; do set -- $cfg
  rm -f coolpuppy_amd/libpup_hip.so coolpuppy_amd/csrc/_obj/pup_engine.o
  COOLPUPPY_AMD_EXTRA_CXXFLAGS="-DPUP_K1S_U=$1 -DPUP_K1S_NA=$2 -DPUP_K1S_WAVES=${3:-4}" python -c "from coolpuppy_amd import build as b; b.build_hip()" > $OUT/build.txt 2>&1
  timeout 400 python tools/probe_trans.py ${CHUNKS:-0,120,250} > $OUT/probe_U$1_NA$2_W${3:-4}.txt 2>&1
done
cat $OUT/tests.txt; for f in $OUT/probe_*.txt; do echo $f; grep chunk $f; done
