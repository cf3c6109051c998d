# scratch batch for one gpurun call (edited per experiment)
timeout -k 10 600 python -m pytest tests/test_gpu_exact.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/a1_tests.log 2>&1; tail -3 gpurun_out/a1_tests.log
HRT_SPP=64 bash tools/variants.sh run cornell_mesh mesh_in_box backrooms_pool random_spheres > gpurun_out/a1_var.log 2>&1; grep Msamples gpurun_out/a1_var.log
