#!/bin/bash
# Timing-only ablations of the fused field kernel (outputs of ablated builds are garbage by construction).
# Separate libraries under /tmp selected with SNERF_LIB; the shipped .so is never touched.
cd "${GRAFT_REPO_ROOT:?set GRAFT_REPO_ROOT (the repo root on the GPU box)}"
D="season-nerf_amd"
for abl in 0 1 2 4 3 7; do
  hipcc -std=c++17 -O3 --offload-arch=gfx950 -fPIC -shared -ffp-contract=off -DSNERF_ABLATE -DABL=$abl -Wno-unused-command-line-argument \
     -o /tmp/abl_$abl.so $D/csrc/kernels.hip $D/csrc/api.cpp $D/csrc/pack.cpp $D/csrc/gemm.hip $D/csrc/train_kernels.hip $D/csrc/train.cpp $D/csrc/dsm.hip &
done
wait
for abl in 0 1 2 4 3 7; do
  echo -n "ABL=$abl (1: no sin/split, 2: no LDS reads, 4: no ring/DMA/barrier): "
  SNERF_LIB=/tmp/abl_$abl.so python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('kernel_ms %.4f' % d['roofline']['kernel_ms'])"
done
