# scratch batch for one gpurun call (edited per experiment)
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/gpu_tests.log 2>&1; tail -5 gpurun_out/gpu_tests.log
