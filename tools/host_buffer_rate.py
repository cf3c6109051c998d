import importlib, sys
sys.path.insert(0, ".")
hrt = importlib.import_module("hai719-raytracing_amd")
hrt.init(0)
w, h, spp = 1920, 1080, 256
s = hrt.HostScene().setup("cornell_mesh", w / h, 1); d = s.flatten(); cam = hrt.default_camera(w / h)
dev = hrt.DeviceScene(d)
dev.render(cam, 64, 64, 1, 1)
for _ in range(3):
    img, st = dev.render(cam, w, h, spp, 1, flags=hrt.FLAG_GAMMA)
    print(f"kernel {st.kernel_ms:.2f} ms  total (launch + gamma + assemble + D2H into caller's buffer) {st.total_ms:.2f} ms -> {w*h*spp/st.total_ms/1e3:.1f} Msamples/s host-to-host vs {w*h*spp/st.kernel_ms/1e3:.1f} kernel only")
