python -m pytest tests/test_gpu_exact.py tests/test_gpu_parity.py tests/test_gpu_kat.py -m gpu -q -x > gpurun_out/t15.log 2>&1; tail -3 gpurun_out/t15.log
tools/variants.sh run cornell_mesh mesh_in_box backrooms_pool 2>&1 | grep Msamples
python tools/time_only.py flamingo_pond raccoon flamingo random_spheres 2>&1 | grep Msamples
