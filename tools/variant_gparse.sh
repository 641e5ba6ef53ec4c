#!/bin/bash
# the shipped library with ONLY hvq_gparse.hip recompiled with other flags -> hvqm4_amd/abl/libhvq_<name>.so
#   tools/variant_gparse.sh <name> <flags replacing -O3 ...>
set -e
name=$1; shift
cd "$(dirname "$0")/../hvqm4_amd/csrc"
make > /dev/null 2>&1
mkdir -p ../abl
hipcc --offload-arch=gfx950 -fPIC -fvisibility=hidden "$@" -c hvq_gparse.hip -o /tmp/g_$name.o
hipcc --offload-arch=gfx950 -shared -fPIC hvq_parse.o hvq_container.o hvq_kernels.o /tmp/g_$name.o hvq_runtime.o -o ../abl/libhvq_$name.so
hipcc --offload-arch=gfx950 -fPIC "$@" -S --cuda-device-only hvq_gparse.hip -o /tmp/g_$name.s 2>/dev/null
echo "built libhvq_$name.so; code bytes: $(grep codeLenInByte /tmp/g_$name.s | awk '{s+=$4} END {print s}')  vgpr: $(grep next_free_vgpr /tmp/g_$name.s | awk '{print $2}' | tr '\n' ' ')  lds: $(grep group_segment_fixed_size /tmp/g_$name.s | awk '{print $2}' | tr '\n' ' ')"
