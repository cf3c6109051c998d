# scratch batch for one gpurun call (edited per experiment)
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/gpu_tests.log 2>&1; tail -4 gpurun_out/gpu_tests.log
python tools/make_frame_means.py > gpurun_out/frame_means.log 2>&1; tail -8 gpurun_out/frame_means.log
HRT_SPP=64 python tools/time_only.py cornell_box cornell_mesh 2>&1 | grep Msamples
HRT_KERNEL=single HRT_SPP=64 python tools/time_only.py cornell_box 2>&1 | grep Msamples
