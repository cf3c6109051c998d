# batch for one gpurun call: the GPU suite, smoke, timing spot check, the profiles of every configuration, the default bench
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/gpu_tests.log 2>&1; tail -4 gpurun_out/gpu_tests.log
python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
HRT_SPP=64 python tools/time_only.py cornell_mesh mesh_in_box backrooms_pool random_spheres 2>&1 | grep Msamples | sed 's/lds=32 leaf=4//; s/1920x1080@64: min//; s/med.*-> */ /; s/vgpr.*//'
bash tools/profile.sh r03 > gpurun_out/profile_r03.log 2>&1; grep -E "failed|==" gpurun_out/profile_r03.log | cut -c1-200
