timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/t31.log 2>&1; tail -5 gpurun_out/t31.log
