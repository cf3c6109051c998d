python -m pytest tests -m gpu -q -x > gpurun_out/t12.log 2>&1; tail -3 gpurun_out/t12.log
python tools/time_only.py cornell_mesh mesh_in_box backrooms_pool random_spheres cornell_box 2>&1 | grep Msamples
echo "--- no exceptions (ablation)"; HRT_ABL_NO_EXCEPTIONS=1 python tools/time_only.py mesh_in_box backrooms_pool 2>&1 | grep Msamples
for leaf in 2 3 6 8; do echo "--- leaf $leaf"; HRT_KD_LEAF=$leaf python tools/time_only.py mesh_in_box backrooms_pool 2>&1 | grep Msamples; done
