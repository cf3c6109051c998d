"""Times the device output stage (finalize, assemble, PPM encode) at 1080p and 4K; prints GB/s of frame bytes moved."""
import importlib, os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hrt = importlib.import_module("hai719-raytracing_amd")
hrt.init(0)
for w, h in ((1920, 1080), (3840, 2160)):
    per = hrt.tiles_total(w, h)
    sums = torch.rand((per, 64, 3), dtype=torch.float32, device="cuda") * 300
    means = torch.empty_like(sums)
    frame = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    cap = 16 * w * h + 64
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    def timed(fn, reps=20):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps): fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    nbytes = w * h * 12
    t = timed(lambda: hrt.finalize_tiles(sums.data_ptr(), per, 256, hrt.FLAG_GAMMA, means.data_ptr(), 0))
    print(f"{w}x{h} finalize(+gamma) {t*1e6:8.1f} us  {2*nbytes/t/1e9:7.1f} GB/s (read+write)")
    t = timed(lambda: hrt.assemble_frame(means.data_ptr(), per, w, h, 1, frame.data_ptr(), 0))
    print(f"{w}x{h} assemble         {t*1e6:8.1f} us  {2*nbytes/t/1e9:7.1f} GB/s (read+write)")
    t = timed(lambda: hrt.encode_ppm(frame.data_ptr(), w, h, 6, out.data_ptr(), cap, 0))
    print(f"{w}x{h} encode P6        {t*1e6:8.1f} us  {(nbytes + w*h*3)/t/1e9:7.1f} GB/s (incl. stream sync)")
    n = 0
    def p3():
        global n
        n = hrt.encode_ppm(frame.data_ptr(), w, h, 3, out.data_ptr(), cap, 0)
    t = timed(p3, 5)
    print(f"{w}x{h} encode P3        {t*1e6:8.1f} us  {n/1e6:.1f} MB of text, {(2*nbytes + n)/t/1e9:7.1f} GB/s (two passes + host scan of block totals)")
    host = frame.cpu().numpy()
    t0 = time.perf_counter(); hrt.write_ppm("/tmp/_t.ppm", host); t1 = time.perf_counter()
    print(f"{w}x{h} host P3 writer (reference's loop)  {1e3*(t1-t0):8.1f} ms")
