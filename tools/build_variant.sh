#!/bin/bash
# usage: tools/build_variant.sh NAME [-DFOO=1 ...]  ->  blub_amd/libblubhip_NAME.so (select with BLUBHIP_LIB=...; A/B measurements of compile-time choices)
name=$1; shift
root=$(cd $(dirname $0)/.. && pwd)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-gpu-rdc -I$root/include -I$root/blub_amd/csrc -Wall -Wno-unused-function "$@" \
  -x hip $root/blub_amd/csrc/blub_fluid.hip $root/blub_amd/csrc/scene_host.cpp $root/blub_amd/csrc/scheduler_host.cpp -o $root/blub_amd/libblubhip_$name.so -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
