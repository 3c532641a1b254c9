#!/bin/bash
# A/B of build-time variants of the field kernel in one process-per-variant run (bench.py, same box).
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}"
D="season-nerf_amd"
i=0
for flags in "$@"; do
  i=$((i+1))
  hipcc -std=c++17 -O3 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off $flags -Wno-unused-command-line-argument \
     -o /tmp/var_$i.so $D/csrc/kernels.hip $D/csrc/api.cpp $D/csrc/pack.cpp $D/csrc/gemm.hip $D/csrc/train_kernels.hip $D/csrc/train.cpp $D/csrc/dsm.hip &
done
wait
for rep in 1 2; do
i=0
for flags in "$@"; do
  i=$((i+1))
  echo -n "[$flags] "
  SNERF_LIB=/tmp/var_$i.so python bench.py --steps 40 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('kernel_ms %.4f' % d['roofline']['kernel_ms'])"
done
done
