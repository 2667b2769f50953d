#!/bin/bash
# Dev tool: build a k_step tuning variant of the (16,0) kernel set into gpurun_out-free scratch: tools/variants/<name>.so
#   tools/build_variant.sh <name> <extra hipcc flags...>
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p tools/variants/obj_$name
for u in me_api me_generic me_statistics; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include "$@" -c metropolisengine_amd/csrc/$u.hip -o tools/variants/obj_$name/$u.o &
done
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I include -DME_NR=16 -DME_NC=0 -DME_DENSE=0 -DME_PER_CHAIN=1 "$@" \
  -c metropolisengine_amd/csrc/me_kernels.hip -o tools/variants/obj_$name/k.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o tools/variants/$name.so tools/variants/obj_$name/*.o
echo built tools/variants/$name.so
