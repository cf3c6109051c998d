# scratch batch for one gpurun call (edited per experiment)
HRT_SPP=64 bash tools/variants.sh run cornell_mesh mesh_in_box backrooms_pool > gpurun_out/a8_var.log 2>&1; grep -E "Msamples|FAILED" gpurun_out/a8_var.log
for leaf in 2 3 6 8; do HRT_KD_LEAF=$leaf HRT_SPP=64 python tools/time_only.py cornell_mesh mesh_in_box backrooms_pool 2>&1 | grep Msamples; done
