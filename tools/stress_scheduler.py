"""Ad-hoc soak of the streaming kernel's scheduler: many launches of varied shapes, each compared with the lane-per-pixel kernel's bits."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hrt = importlib.import_module("hai719-raytracing_amd")
hrt.init(0)
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cases = [("cornell_mesh", 320, 180, 6), ("backrooms_pool", 200, 112, 3), ("random_spheres", 160, 90, 5), ("cornell_mesh", 64, 36, 130),
         ("mesh_in_box", 97, 61, 17), ("flamingo", 80, 45, 4), ("cornell_box", 24, 16, 300), ("cornell_mesh", 640, 360, 2)]
t0 = time.time(); n = 0
for name, w, h, spp in cases:
    dev = hrt.DeviceScene(hrt.HostScene().setup(name, w / h, 1).flatten()); cam = hrt.default_camera(w / h)
    want, _ = dev.render(cam, w, h, spp, seed=5, flags=hrt.FLAG_WAVE_KERNEL)
    for k in range(rounds):
        got, _ = dev.render(cam, w, h, spp, seed=5, flags=hrt.FLAG_STREAM_KERNEL)
        assert np.array_equal(got, want), (name, w, h, spp, k)
        n += 1
    print(f"{name} {w}x{h}@{spp}: {rounds} launches identical", flush=True)
print(f"{n} launches in {time.time() - t0:.1f} s: all identical to the lane-per-pixel kernel")
