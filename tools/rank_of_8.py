"""One rank's launch of an 8-rank weak-scaling step on one GPU: tiles rank, rank + 8, ... of a 1080p frame at 8 x 256 spp."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hrt = importlib.import_module("hai719-raytracing_amd")
hrt.init(0)
W, H, world = 1920, 1080, 8
rank = int(sys.argv[1]) if len(sys.argv) > 1 else 3
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
scene = hrt.DeviceScene(hrt.HostScene().setup("cornell_mesh", W / H, 1).flatten()); cam = hrt.default_camera(W / H)
tiles = ((W + 7) // 8) * ((H + 7) // 8)
owned = (tiles - rank + world - 1) // world
buf = torch.zeros(owned * 192, dtype=torch.float32, device="cuda")
stream = torch.cuda.current_stream().cuda_stream
for k in range(2):
    t0 = time.perf_counter()
    scene.render_tiles(cam, W, H, spp, 1, 0, rank, world, buf.data_ptr(), stream)
    torch.cuda.synchronize()
    print(f"rank {rank}/{world} spp {spp}: {1e3 * (time.perf_counter() - t0):.1f} ms, kernel {scene.last_kernel_ms():.1f} ms, mean {float(buf.mean()):.6f}, finite {bool(torch.isfinite(buf).all())}")
scene.check_last_launch() if hasattr(scene, "check_last_launch") else None
