# scratch batch for one gpurun call (edited per experiment)
cd hai719-raytracing_amd
HRT_LIBNAME=libhrt_var_dbg.so python3 ../tools/sp_debug_report.py random_spheres:64 cornell_mesh:64 backrooms_pool:32 mesh_in_box:64
for k in 3 4 1; do HRT_SEG_KIND=$k HRT_LIBNAME=libhrt_var_seg$k.so python3 ../tools/seg_report.py random_spheres 32; done
for k in 3 4; do HRT_SEG_KIND=$k HRT_LIBNAME=libhrt_var_seg${k}n.so python3 ../tools/seg_report.py random_spheres 32; done
HRT_SEG_KIND=1 HRT_LIBNAME=libhrt_var_seg1.so python3 ../tools/seg_report.py cornell_mesh 32
HRT_SEG_KIND=1 HRT_LIBNAME=libhrt_var_seg1.so python3 ../tools/seg_report.py backrooms_pool 32
