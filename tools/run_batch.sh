FUZZ_ORACLE=1 FUZZ_MAX_ENTRIES=20000 timeout -k 10 1000 python tools/fuzz_exact.py 1500 40000 > gpurun_out/fuzz5.log 2>&1; tail -4 gpurun_out/fuzz5.log; grep -c identical gpurun_out/fuzz5.log
