#!/bin/bash
# build a tuning / diagnostic variant of the whole library:
#   tools/variant.sh <name> <extra -D flags...>  -> hvqm4_amd/abl/libhvq_<name>.so   (select it with HVQM4_AMD_LIB)
set -e
name=$1; shift
cd "$(dirname "$0")/../hvqm4_amd/csrc"
mkdir -p ../abl /tmp/var_$name
gcc -O3 -march=x86-64-v3 -fPIC -Wall -Wextra "$@" -c hvq_parse.c -o /tmp/var_$name/p.o
gcc -O2 -fPIC -Wall -Wextra "$@" -c hvq_container.c -o /tmp/var_$name/c.o
hipcc --offload-arch=gfx950 -O3 -fPIC -fvisibility=hidden -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -amdgpu-kernarg-preload-count=4 "$@" -c hvq_kernels.hip -o /tmp/var_$name/k.o
hipcc --offload-arch=gfx950 -O3 -fPIC -fvisibility=hidden "$@" -c hvq_gparse.hip -o /tmp/var_$name/g.o
hipcc --offload-arch=gfx950 -O2 -fPIC -fvisibility=hidden "$@" -c hvq_runtime.cpp -o /tmp/var_$name/r.o
hipcc --offload-arch=gfx950 -shared -fPIC /tmp/var_$name/p.o /tmp/var_$name/c.o /tmp/var_$name/k.o /tmp/var_$name/g.o /tmp/var_$name/r.o -o ../abl/libhvq_$name.so
echo "built hvqm4_amd/abl/libhvq_$name.so"
