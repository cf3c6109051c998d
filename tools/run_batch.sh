HRT_SPP=64 tools/variants.sh run backrooms_pool mesh_in_box cornell_mesh 2>&1 | grep "Msamples\|FAILED\|rror"
