python -m pytest tests/test_gpu_exact.py tests/test_gpu_parity.py tests/test_gpu_kat.py -m gpu -q -x > gpurun_out/t17.log 2>&1; tail -4 gpurun_out/t17.log
python tools/time_only.py cornell_mesh mesh_in_box backrooms_pool random_spheres cornell_box 2>&1 | grep Msamples
