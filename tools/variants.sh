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
    make -s -B libhrt_var_$name.so VARIANT=libhrt_var_$name.so VARIANT_FLAGS="$flags" || exit 1   # the Makefile's two-pass build (no packed fp32 in device code)
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
