"""Output stage on the device: progressive accumulation / resume and the PPM encoders.
Everything here is bit-exact: the progressive path must leave the bits of the one-shot render, and the
encoded file must be the bytes the reference's `ofstream <<` loop writes (main.cpp:258-262)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def build(gpu, name, aspect):
    host = gpu.HostScene().setup(name, aspect, 1)
    desc = host.flatten()
    return desc, gpu.DeviceScene(desc), gpu.default_camera(aspect)


def one_shot(gpu, dev, cam, w, h, spp, seed, flags, world=1):
    import torch
    per = gpu.tiles_owned(w, h, 0, world)
    gathered = torch.zeros((world, per, 64, 3), dtype=torch.float32, device="cuda")
    for r in range(world):
        dev.render_tiles(cam, w, h, spp, seed, flags, r, world, gathered[r].data_ptr(), 0)
    frame = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    gpu.assemble_frame(gathered.data_ptr(), per, w, h, world, frame.data_ptr(), 0)
    torch.cuda.synchronize()
    return frame


@pytest.mark.parametrize("name,flags_name,world", [("cornell_mesh", None, 1), ("cornell_mesh", "FLAG_WAVE_KERNEL", 2), ("cornell_mesh", "FLAG_DUAL_KERNEL", 1),
                                                   ("random_spheres", None, 1), ("backrooms_pool", None, 3)])
def test_progressive_chunks_leave_the_bits_of_the_one_shot_render(gpu, name, flags_name, world):
    """Samples added in chunks of 3 + 1 + 8 + 4 (per-pixel sums continue in sample order) == 16 spp at once,
    with and without gamma, for both lane-per-pixel kernel forms and across tile partitions."""
    import torch
    w, h, seed = 120, 67, 17
    flags = getattr(gpu, flags_name) if flags_name else 0
    desc, dev, cam = build(gpu, name, w / h)
    per = gpu.tiles_owned(w, h, 0, world)
    sums = torch.zeros((world, per, 64, 3), dtype=torch.float32, device="cuda")
    first = 0
    for n in (3, 1, 8, 4):
        for r in range(world):
            dev.render_accumulate(cam, w, h, first, n, seed, flags, r, world, sums[r].data_ptr(), 0)
        first += n
    for gamma in (0, gpu.FLAG_GAMMA):
        means = torch.empty_like(sums)
        gpu.finalize_tiles(sums.data_ptr(), world * per, first, gamma, means.data_ptr(), 0)
        frame = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
        gpu.assemble_frame(means.data_ptr(), per, w, h, world, frame.data_ptr(), 0)
        torch.cuda.synchronize()
        ref = one_shot(gpu, dev, cam, w, h, first, seed, flags | gamma, world)
        assert float(ref.max()) > 0
        assert torch.equal(frame, ref)


def test_resume_from_a_checkpoint_on_a_fresh_scene_object(gpu):
    """Checkpoint = the sum buffer + the number of samples done.  A new process (here: a new DeviceScene)
    that reloads it and continues ends with the same bits as an uninterrupted render."""
    import torch
    w, h, seed = 64, 40, 3
    desc, dev, cam = build(gpu, "mesh_in_box", w / h)
    per = gpu.tiles_owned(w, h, 0, 1)
    sums = torch.zeros((per, 64, 3), dtype=torch.float32, device="cuda")
    dev.render_accumulate(cam, w, h, 0, 5, seed, 0, 0, 1, sums.data_ptr(), 0)
    torch.cuda.synchronize()
    checkpoint = sums.cpu().numpy().copy()
    del dev
    desc2, dev2, cam2 = build(gpu, "mesh_in_box", w / h)
    sums2 = torch.from_numpy(checkpoint).cuda()
    dev2.render_accumulate(cam2, w, h, 5, 6, seed, 0, 0, 1, sums2.data_ptr(), 0)
    gpu.finalize_tiles(sums2.data_ptr(), per, 11, 0, sums2.data_ptr(), 0)  # in place
    frame = torch.empty((h, w, 3), dtype=torch.float32, device="cuda")
    gpu.assemble_frame(sums2.data_ptr(), per, w, h, 1, frame.data_ptr(), 0)
    torch.cuda.synchronize()
    ref, _ = dev2.render(cam2, w, h, 11, seed=seed)
    assert np.array_equal(frame.cpu().numpy(), ref)


def encode(gpu, frame_t, fmt):
    import torch
    h, w, _ = frame_t.shape
    cap = 40 * w * h + 64
    out = torch.zeros(cap, dtype=torch.uint8, device="cuda")
    n = gpu.encode_ppm(frame_t.data_ptr(), w, h, fmt, out.data_ptr(), cap, 0)
    return bytes(out[:n].cpu().numpy())


def test_ppm_encoded_on_the_device_is_the_reference_file_byte_for_byte(gpu):
    import torch
    # a rendered frame (gamma-corrected, as the reference writes it) ...
    w, h = 203, 77  # not a multiple of the 1024-pixel encoder block
    desc, dev, cam = build(gpu, "cornell_box", w / h)
    frame = one_shot(gpu, dev, cam, w, h, 4, 9, gpu.FLAG_GAMMA)
    host = frame.cpu().numpy()
    assert encode(gpu, frame, 3) == gpu.ppm_text_reference(host)
    p6 = encode(gpu, frame, 6)
    head = f"P6\n{w} {h}\n255\n".encode()
    want = np.trunc(np.float32(255.0) * np.minimum(host, np.float32(1.0))).astype(np.uint8).tobytes()
    assert p6 == head + want
    # ... and the values a renderer should never produce but the reference's expression still defines:
    # > 1 clamps, NaN -> 255, negatives print with a sign (P3) / clamp to 0 (P6), -inf -> INT_MIN
    rng = np.random.default_rng(5)
    odd = rng.uniform(-0.2, 1.3, size=(31, 45, 3)).astype(np.float32)
    odd[0, 0] = (np.nan, -np.inf, np.inf)
    odd[1, 1] = (-1e30, 1e30, -0.0)
    odd[2, 2] = (0.999999, 1.0 / 255.0, 0.00392156)
    t = torch.from_numpy(odd).cuda()
    assert encode(gpu, t, 3) == gpu.ppm_text_reference(odd)
    p6 = encode(gpu, t, 6)
    m = np.where(odd < 1, odd, np.float32(1.0))
    v = np.float32(255.0) * m
    want = np.where(np.isfinite(v), np.clip(np.trunc(v), 0, 255), 0).astype(np.uint8).tobytes()
    assert p6 == f"P6\n45 31\n255\n".encode() + want


def test_encoder_rejects_a_buffer_that_is_too_small(gpu):
    import torch
    t = torch.full((4, 4, 3), 0.5, dtype=torch.float32, device="cuda")
    out = torch.zeros(16, dtype=torch.uint8, device="cuda")
    with pytest.raises(gpu.HrtError, match="too small"):
        gpu.encode_ppm(t.data_ptr(), 4, 4, 3, out.data_ptr(), 16, 0)
    with pytest.raises(gpu.HrtError):
        gpu.encode_ppm(t.data_ptr(), 4, 4, 5, out.data_ptr(), 16, 0)


def test_raytracer_driver_writes_the_file_the_library_renders(gpu, tmp_path):
    """host/raytracer_main.cpp is the reference's `'r'` key without GLUT: scene set-up, flatten,
    hrt_render with gamma, P3 dump.  Its file must be the bytes of the same render through the Python binding."""
    import os
    import subprocess
    from conftest import PKG, ROOT
    w, h, spp, seed = 96, 54, 3, 5
    out = tmp_path / "rendu.ppm"
    exe = os.path.join(PKG, "raytracer")
    r = subprocess.run([exe, "--scene", "cornell_mesh", "--w", str(w), "--h", str(h), "--spp", str(spp), "--seed", str(seed),
                        "--assets", os.path.join(ROOT, "assets"), "--out", str(out)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    desc, dev, cam = build(gpu, "cornell_mesh", w / h)
    img, _ = dev.render(cam, w, h, spp, seed=seed, flags=gpu.FLAG_GAMMA)
    assert out.read_bytes() == gpu.ppm_text_reference(img)


def test_a_launch_that_gives_up_is_reported_not_returned(gpu, tmp_path):
    """The streaming kernel's scheduler has a bound on the cycles it may spin with nothing to run, so that a bug can never spin
    the GPU.  libhrt_var_bound.so is the same library built to give up after 3 serial sections (Makefile): every entry point that hands back pixels must then fail
    with HRT_ERR_DEVICE -- hrt_render (with and without stats), hrt_check_last_launch after the asynchronous entry points,
    hrt_last_kernel_ms -- and the process must stay usable (the lane-per-pixel kernel has no such bound and still renders)."""
    import subprocess, sys, textwrap
    code = textwrap.dedent('''
        import ctypes as C, importlib, sys
        import numpy as np
        sys.path.insert(0, %r)
        hrt = importlib.import_module("hai719-raytracing_amd")
        import torch
        hrt.init(0)
        w, h = 64, 48
        host = hrt.HostScene().setup("cornell_mesh", w / h, 1); desc = host.flatten(); cam = hrt.default_camera(w / h)
        dev = hrt.DeviceScene(desc)
        lib = hrt.device_lib()
        out = np.empty((h, w, 3), np.float32)
        rc1 = lib.hrt_render(dev._h, C.byref(cam), w, h, 8, 1, hrt.FLAG_STREAM_KERNEL, out.ctypes.data, None)
        st = hrt.Stats()
        rc2 = lib.hrt_render(dev._h, C.byref(cam), w, h, 8, 1, hrt.FLAG_STREAM_KERNEL, out.ctypes.data, C.byref(st))
        msg = lib.hrt_last_error().decode()
        tiles = torch.zeros((hrt.tiles_total(w, h), 64, 3), dtype=torch.float32, device="cuda")
        dev.render_tiles(cam, w, h, 8, 1, hrt.FLAG_STREAM_KERNEL, 0, 1, tiles.data_ptr(), 0)
        rc3 = lib.hrt_check_last_launch(dev._h)
        ms = C.c_double()
        rc4 = lib.hrt_last_kernel_ms(dev._h, C.byref(ms))
        img, _ = dev.render(cam, w, h, 2, 1, flags=hrt.FLAG_WAVE_KERNEL)   # still alive
        print("RC", rc1, rc2, rc3, rc4, float(img.max()) > 0, "|", msg)
    ''') % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))),)
    env = dict(os.environ, HRT_LIBNAME="libhrt_var_bound.so")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("RC")][0]
    assert line.startswith("RC -2 -2 -2 -2 True |") and "gave up" in line, line


@pytest.mark.parametrize("name,w,h,spp", [("cornell_mesh", 200, 120, 3), ("random_spheres", 61, 35, 2), ("backrooms_pool", 128, 72, 2)])
def test_render_multi_gives_the_bits_of_render(gpu, name, w, h, spp):
    """hrt_multi_* / hrt_render_multi: one process, one replica + stream per slot, tiles r, r + N, ..., ONE gather of the
    tile buffers to slot 0, assemble.  With every slot on GPU 0 (an ordinal may repeat) the path runs on a one-GPU box:
    1, 2, 3 and 5 slots must each give exactly the frame hrt_render gives, with and without gamma."""
    desc, dev, cam = build(gpu, name, w / h)
    want, _ = dev.render(cam, w, h, spp, seed=9)
    want_g, _ = dev.render(cam, w, h, spp, seed=9, flags=gpu.FLAG_GAMMA)
    for slots in ([0], [0, 0], [0, 0, 0], [0] * 5):
        ms = gpu.MultiScene(desc, slots)
        # one slot = distinct ordinals: the gather of one rank goes through an RCCL communicator (ncclCommInitAll + ncclGather);
        # a repeated ordinal cannot (a communicator does not take a device twice): peer copies
        assert ms.gather == ("rccl" if len(slots) == 1 else "peer"), (slots, ms.gather, ms.note)
        got, st = ms.render(cam, w, h, spp, seed=9)
        assert np.array_equal(got, want), f"{len(slots)} slots"
        assert st.samples == w * h * spp and st.kernel_ms > 0
        again, _ = ms.render(cam, w, h, spp, seed=9, flags=gpu.FLAG_GAMMA)   # buffers reused
        assert np.array_equal(again, want_g)
        ms.close()
    one_shot, _ = gpu.render_multi(desc, cam, w, h, spp, 9, 0, [0, 0, 0])
    assert np.array_equal(one_shot, want)
    bigger, _ = gpu.render_multi(desc, cam, 3 * w, 2 * h, 1, 9, 0, [0, 0])
    assert np.array_equal(bigger, dev.render(cam, 3 * w, 2 * h, 1, seed=9)[0])
    with pytest.raises(gpu.HrtError):
        gpu.MultiScene(desc, [0, 99])          # no such device
    img, _ = dev.render(cam, w, h, spp, seed=9)  # the single-GPU scene still works afterwards
    assert np.array_equal(img, want)


def test_multi_gather_can_be_chosen_and_is_reported(gpu, monkeypatch, tmp_path):
    """HRT_MULTI_GATHER=peer forces the peer copies even where a communicator exists; =rccl refuses a repeated ordinal instead
    of falling back; the driver prints which gather ran.  Both forms give hrt_render's bits."""
    import subprocess
    w, h, spp = 96, 54, 2
    desc, dev, cam = build(gpu, "cornell_mesh", w / h)
    want, _ = dev.render(cam, w, h, spp, seed=3)
    monkeypatch.setenv("HRT_MULTI_GATHER", "peer")
    ms = gpu.MultiScene(desc, [0])
    assert ms.gather == "peer" and np.array_equal(ms.render(cam, w, h, spp, seed=3)[0], want)
    ms.close()
    monkeypatch.setenv("HRT_MULTI_GATHER", "rccl")
    ms = gpu.MultiScene(desc, [0])
    assert ms.gather == "rccl" and ms.note == ""
    assert np.array_equal(ms.render(cam, w, h, spp, seed=3)[0], want)
    assert np.array_equal(ms.render(cam, 2 * w, 2 * h, 1, seed=3)[0], dev.render(cam, 2 * w, 2 * h, 1, seed=3)[0])  # buffers regrown
    ms.close()
    with pytest.raises(gpu.HrtError, match="distinct"):
        gpu.MultiScene(desc, [0, 0])
    monkeypatch.setenv("HRT_MULTI_GATHER", "carrier pigeon")
    with pytest.raises(gpu.HrtError, match="rccl or peer"):
        gpu.MultiScene(desc, [0])
    monkeypatch.delenv("HRT_MULTI_GATHER")
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hai719-raytracing_amd")
    for extra, name in ((["--gpus", "1"], "rccl"), (["--devices", "0,0"], "peer")):
        r = subprocess.run([os.path.join(pkg, "raytracer"), "--scene", "cornell_box", "--w", "32", "--h", "32", "--spp", "1", "--assets",
                            os.path.join(os.path.dirname(pkg), "assets"), "--out", os.path.join(str(tmp_path), "g.ppm")] + extra,
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and f"gather: {name}" in r.stdout, r.stdout + r.stderr


def test_raytracer_driver_renders_across_slots(gpu, tmp_path):
    """The headless driver (main.cpp without GLUT): --devices 0,0 writes the same PPM as the single-GPU run."""
    import subprocess
    pkg = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hai719-raytracing_amd")
    outs = []
    for extra in ([], ["--devices", "0,0"]):
        out = os.path.join(str(tmp_path), f"r{len(outs)}.ppm")
        r = subprocess.run([os.path.join(pkg, "raytracer"), "--scene", "cornell_mesh", "--w", "96", "--h", "54", "--spp", "3", "--seed", "4",
                            "--assets", os.path.join(os.path.dirname(pkg), "assets"), "--out", out] + extra, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        outs.append(open(out, "rb").read())
    assert outs[0] == outs[1] and outs[0].startswith(b"P3\n96 54\n255\n")
