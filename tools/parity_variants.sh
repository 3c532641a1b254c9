#!/bin/bash
# run the network parity test against several build variants (SNERF_LIB)
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}"
D="season-nerf_amd"
i=0
for flags in "$@"; do
  i=$((i+1))
  hipcc -std=c++17 -O3 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off $flags -Wno-unused-command-line-argument \
     -o /tmp/pvar_$i.so $D/csrc/kernels.hip $D/csrc/api.cpp $D/csrc/pack.cpp $D/csrc/gemm.hip $D/csrc/train_kernels.hip $D/csrc/train.cpp $D/csrc/dsm.hip &
done
wait
i=0
for flags in "$@"; do
  i=$((i+1))
  echo "== [$flags]"
  SNERF_LIB=/tmp/pvar_$i.so python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "network" 2>&1 | grep "fwd_Rho\|fwd_Col \|passed\|failed"
done
