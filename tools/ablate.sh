#!/bin/bash
# Build ablation variants of the kernel (not the product) into gpurun_out/abl/ -- run on the build box.
set -e
cd "$(dirname "$0")/../hvqm4_amd/csrc"
mkdir -p ../abl
make -s hvq_parse.o hvq_runtime.o
for v in "$@"; do
  hipcc --offload-arch=gfx950 -O3 -fPIC -fvisibility=hidden -DHVQ_ABL=$v -c hvq_kernels.hip -o /tmp/hvq_kernels_abl$v.o
  hipcc --offload-arch=gfx950 -shared -fPIC hvq_parse.o /tmp/hvq_kernels_abl$v.o hvq_runtime.o -o ../abl/libhvq_abl$v.so
done
