# batch for one gpurun call: the GPU suite, smoke, the profiles of every configuration, the default bench
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > gpurun_out/gpu_tests.log 2>&1; tail -4 gpurun_out/gpu_tests.log
python __graft_entry__.py --smoke > gpurun_out/smoke.log 2>&1; tail -1 gpurun_out/smoke.log
bash tools/profile.sh r03 > gpurun_out/profile_r03.log 2>&1; grep -E "failed|==" gpurun_out/profile_r03.log | cut -c1-200
cp gpurun_out/prof_r03/r03_*_pmc.json gpurun_out/prof_r03/r03_*_kernel_stats.csv profiles/   # (this copy of the tree is scratch; the same files are copied into profiles/ at home)
python bench.py > gpurun_out/bench_r03.log 2>&1; tail -c 600 gpurun_out/bench_r03.log
