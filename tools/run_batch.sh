# scratch batch for one gpurun call (edited per experiment)
timeout -k 10 900 python -m pytest tests -m gpu -q -x > gpurun_out/gpu_tests.log 2>&1; tail -5 gpurun_out/gpu_tests.log
HRT_SPP=64 python tools/time_only.py random_spheres rt_in_a_weekend flamingo mesh_in_box 2>&1 | grep Msamples
