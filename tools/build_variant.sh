#!/bin/bash
# Dev tool: build a tuning variant of ONE kernel set (default (16,0)) into scratch: tools/variants/<name>.so
#   [NR=64 NC=0 DENSE=1 PER_CHAIN=0] tools/build_variant.sh <name> <extra hipcc flags...>
# and select it with METROPOLIS_HIP_LIB=tools/variants/<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/variants/obj_$name
for u in me_api me_generic me_statistics me_runtime_dims; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include "$@" -c metropolisengine_amd/csrc/$u.hip -o tools/variants/obj_$name/$u.o &
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -DME_NR=${NR:-16} -DME_NC=${NC:-0} -DME_DENSE=${DENSE:-0} -DME_PER_CHAIN=${PER_CHAIN:-1} "$@" \
  -c metropolisengine_amd/csrc/me_kernels.hip -o tools/variants/obj_$name/k.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/$name.so tools/variants/obj_$name/*.o
echo built tools/variants/$name.so
