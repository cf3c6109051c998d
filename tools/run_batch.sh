# scratch batch for one gpurun call (edited per experiment): profile + bench of the build in the tree
bash tools/profile.sh r02 > gpurun_out/profile_r02.log 2>&1; tail -14 gpurun_out/profile_r02.log
cp gpurun_out/prof_r02/r02_pmc.json profiles/r02_pmc.json
python bench.py --steps 5 --warmup 1 > gpurun_out/bench_r02h.log 2>&1; tail -c 300 gpurun_out/bench_r02h.log
