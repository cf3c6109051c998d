import importlib, os, sys
sys.path.insert(0, "/root/repo")
hrt = importlib.import_module("hai719-raytracing_amd")
hrt.init(0)
dev = hrt.DeviceScene(hrt.HostScene().setup("cornell_box", 1.0, 1).flatten()); cam = hrt.default_camera(1.0)
for name, fl in (("auto", 0), ("lane", hrt.FLAG_WAVE_KERNEL), ("stream", hrt.FLAG_STREAM_KERNEL)):
    for w, h, spp in ((256, 256, 4), (256, 256, 64), (1920, 1080, 16)):
        ms = []
        for _ in range(6):
            _, st = dev.render(cam, w, h, spp, 1, flags=fl); ms.append(st.kernel_ms)
        print(f"{name:7s} {w}x{h}@{spp}: {min(ms):.3f} ms -> {w*h*spp/min(ms)/1e3:.0f} Msamples/s")
