python tools/fuzz_debug.py 1011 2>&1 | grep differ
HRT_LIBNAME=libhrt_var_p0.so python tools/fuzz_debug.py 1011 2>&1 | grep differ
