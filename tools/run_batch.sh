# scratch batch for one gpurun call (edited per experiment): the full GPU suite
timeout -k 10 900 python -m pytest tests -m gpu -q > gpurun_out/gpu_tests.log 2>&1; tail -5 gpurun_out/gpu_tests.log
