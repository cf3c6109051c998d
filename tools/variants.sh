#!/bin/bash
# Build A/B variants of libhrt.so (same ABI, different -D flags) in the build container:
#   tools/variants.sh build "name1:-DFLAG=1" "name2:-DFLAG=2" ...
# and time them on the GPU box (one process per variant):
#   tools/variants.sh run [scene ...]
cd "$(dirname "$0")/../hai719-raytracing_amd"
if [ "$1" = "build" ]; then
  shift
  for f in libhrt_var_*.so; do [ "$f" = libhrt_var_bound.so ] || rm -f "$f"; done
  for spec in "$@"; do
    name=${spec%%:*}; flags=${spec#*:}
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off -fvisibility=hidden --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-bitwise-instead-of-logical $flags -shared -o libhrt_var_$name.so csrc/hrt_api.hip -ldl || exit 1
    echo "built libhrt_var_$name.so ($flags)"
  done
else
  shift
  for lib in libhrt.so libhrt_var_*.so; do
    [ -f $lib ] || continue
    [ $lib = libhrt_var_bound.so ] && continue
    HRT_LIBNAME=$lib timeout -k 10 300 python3 ../tools/time_only.py "$@" || echo "$lib FAILED"
  done
fi
