timeout -k 10 1000 python tools/fuzz_exact.py 6000 2680 > gpurun_out/fuzz3.log 2>&1; tail -3 gpurun_out/fuzz3.log; grep -c identical gpurun_out/fuzz3.log; grep -c "tie between" gpurun_out/fuzz3.log
