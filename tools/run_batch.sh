python -m pytest tests/test_gpu_exact.py tests/test_gpu_parity.py -m gpu -q -x > gpurun_out/t14.log 2>&1; tail -3 gpurun_out/t14.log
python tools/time_only.py cornell_mesh mesh_in_box backrooms_pool flamingo_pond raccoon flamingo 2>&1 | grep Msamples
