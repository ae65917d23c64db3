#!/bin/bash
# exit-abort hunt, round 2.  A: default under gdb (backtrace); B: no prefetch thread; C: torch imported first; D: system HIP
out=gpurun_out/bx
K='not baseline_configs'
rocgdb -batch -ex "set pagination off" -ex run -ex bt --args python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "$K" > ${out}_A_gdb.log 2>&1
grep -n "^#[0-9]\|passed\|failed\|SIGABRT\|exited" ${out}_A_gdb.log | head -60
COOLPUPPY_AMD_NO_PREFETCH=1 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "$K" > ${out}_B.log 2>&1; echo "B noprefetch rc=$?"
python3 -c "import torch, pytest, sys; sys.exit(pytest.main(['tests/','-x','-q','-m','gpu','-p','no:cacheprovider','-k','$K']))" > ${out}_C.log 2>&1; echo "C torchfirst rc=$?"
COOLPUPPY_AMD_SYSTEM_HIP=1 python3 -m pytest tests/ -x -q -m gpu -p no:cacheprovider -k "$K" > ${out}_D.log 2>&1; echo "D systemhip rc=$?"
tail -3 ${out}_B.log ${out}_C.log ${out}_D.log
