"""Randomised check of everything AROUND the trace kernel (GPU box):  python tools/fuzz_plumbing.py [scenes] [first seed]

For random scenes and frame shapes (tools/fuzz_exact.py's generator): the frame rendered in one go must equal, bit for bit,
  * its tiles rendered rank by rank for a random world size and assembled (hrt_render_tiles + hrt_assemble_frame),
  * the same samples accumulated in random chunks and finalised (hrt_render_accumulate + hrt_finalize_tiles),
  * hrt_multi_render over 1-4 slots of GPU 0 (one slot: the RCCL communicator path; more: peer copies),
and the KD-trees built on the device (hrt_kd_build_gpu) must be the host builder's, for random leaf_max / max_depth."""
import ctypes as C, importlib, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
hrt = importlib.import_module("hai719-raytracing_amd")
hrt.init(0)
from test_host_layer import MeshDesc, SceneDesc
src = open(os.path.join(ROOT, "tools", "fuzz_exact.py")).read()
exec(src[src.index("M = hrt.Material.make"):src.index("def host_meshes")])   # material(), scene()
n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1


def trees(desc):
    d = C.cast(desc, C.POINTER(SceneDesc)).contents
    out = []
    for m in range(d.n_meshes):
        mesh = C.cast(d.meshes, C.POINTER(MeshDesc))[m]
        units = np.ctypeslib.as_array(C.cast(mesh.kd_units, C.POINTER(C.c_uint32)), shape=(mesh.n_kd_units, 4)).copy() if mesh.n_kd_units else np.zeros((0, 4), np.uint32)
        leaf = np.ctypeslib.as_array(C.cast(mesh.leaf_tris, C.POINTER(C.c_uint32)), shape=(mesh.n_leaf_tris,)).copy() if mesh.n_leaf_tris else np.zeros(0, np.uint32)
        out.append((units, leaf, int(mesh.kd_root)))
    return out


done = refused = 0
for k in range(n_scenes):
    seed = seed0 + k
    rng = np.random.default_rng(seed + 12345)
    host, ns, nq = scene(seed)
    host.set_kd_params(leaf_max=int(rng.choice([1, 2, 4, 9])), max_depth=int(rng.choice([0, 0, 3, 12])))
    try:
        desc = host.flatten()
    except hrt.HrtError:
        refused += 1
        continue
    if sum(host.irregular_stats(m)["entries"] for m in range(len(trees(desc)))) > 100000:
        continue
    host_trees = trees(desc)
    host.set_kd_builder("gpu")
    gpu_trees = trees(host.flatten())
    host.set_kd_builder(None)
    desc = host.flatten()
    for m, (a, b) in enumerate(zip(host_trees, gpu_trees)):
        assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2], f"seed {seed}: mesh {m}: the GPU-built tree differs from the host builder's"
    w, h = int(rng.integers(1, 300)), int(rng.integers(1, 200))
    spp = int(rng.choice([1, 2, 3, 9, 40]))
    flags = int(rng.choice([0, 0, hrt.FLAG_WAVE_KERNEL, hrt.FLAG_GAMMA]))
    dev = hrt.DeviceScene(desc); cam = hrt.default_camera(w / h)
    want, _ = dev.render(cam, w, h, spp, seed=seed, flags=flags)
    # tiles of a random world, rank by rank
    world = int(rng.choice([2, 3, 5, 8]))
    per = hrt.tiles_owned(w, h, 0, world)
    gathered = torch.zeros((world, max(per, 1), 64, 3), dtype=torch.float32, device="cuda")
    for r in range(world):
        if hrt.tiles_owned(w, h, r, world):
            dev.render_tiles(cam, w, h, spp, seed, flags, r, world, gathered[r].data_ptr(), 0)
    frame = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    hrt.assemble_frame(gathered.data_ptr(), max(per, 1), w, h, world, frame.data_ptr(), 0)
    torch.cuda.synchronize()
    assert np.array_equal(frame.cpu().numpy(), want), f"seed {seed}: tiles of world {world} differ from the frame"
    # the same samples in random chunks
    tiles = hrt.tiles_total(w, h)
    sums = torch.zeros((tiles, 64, 3), dtype=torch.float32, device="cuda")
    s0 = 0
    while s0 < spp:
        n = int(rng.integers(1, spp - s0 + 1))
        dev.render_accumulate(cam, w, h, s0, n, seed, flags & ~hrt.FLAG_GAMMA, 0, 1, sums.data_ptr(), 0)
        s0 += n
    out_tiles = torch.empty_like(sums)
    torch.cuda.synchronize()
    lib = hrt.device_lib()
    assert lib.hrt_finalize_tiles(C.c_void_p(sums.data_ptr()), tiles, spp, flags & hrt.FLAG_GAMMA, C.c_void_p(out_tiles.data_ptr()), None) == 0
    hrt.assemble_frame(out_tiles.data_ptr(), tiles, w, h, 1, frame.data_ptr(), 0)
    torch.cuda.synchronize()
    assert np.array_equal(frame.cpu().numpy(), want), f"seed {seed}: accumulated chunks differ from the one-shot frame"
    # several slots of one process
    slots = [0] * int(rng.integers(1, 5))
    ms = hrt.MultiScene(desc, slots)
    got, _ = ms.render(cam, w, h, spp, seed=seed, flags=flags)
    ms.close()
    assert np.array_equal(got, want), f"seed {seed}: hrt_multi_render over {len(slots)} slots differs"
    dev.close()
    done += 1
    if done % 25 == 0:
        print(f"{done} scenes: tiles / chunks / slots / GPU-built trees identical (last: seed {seed}, {w}x{h}@{spp}, world {world}, {len(slots)} slots)", flush=True)
print(f"{n_scenes} random scenes ({refused} refused by the host layer), {done} checked: all identical")
