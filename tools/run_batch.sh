timeout -k 10 600 python -m pytest tests/test_gpu_kdbuild.py -m gpu -q -x -s > gpurun_out/kd_tests.log 2>&1; tail -25 gpurun_out/kd_tests.log
