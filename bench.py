#!/usr/bin/env python3
"""Headline benchmark of the MI355X ray-trace path.

Metric (BASELINE.json): Msamples/s = pixels x spp / wall-second on Cornell box + flamingo_lowpoly
mesh, 1920x1080 @ 256 spp.  A "step" is one full pass of the hot path over the frame: every rank
renders its share of the 8x8 tiles (hrt_render_tiles), rank 0 gathers the tiles (ONE collective)
and de-interleaves them into the frame.  The scene is resident in HBM before the timed region.

Weak scaling over N GPUs: the frame stays 1920x1080 and is tile-partitioned across the ranks;
samples per pixel grow as 256*N so every GPU traces the same 530.8 M samples as the 1-GPU run.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

(HRT_BENCH_SHARE_GPU=1 rehearses the N > 1 path on a one-GPU box: all ranks share GPU 0, collectives over gloo.)

Rank 0 prints ONE JSON line.  `roofline` prices the trace kernel's algorithmic bytes (SURVEY.md
8(d) record sizes x committed per-sample work counters) against the 8 TB/s HBM peak, with the
kernel's launch duration measured by HIP events on the launch stream.  `cpu_baseline` is the CPU
oracle (a port of the reference algorithm, reference-shaped KD-tree) timed on this host on a
bounded sample of the same workload -- a reported baseline, not the target.
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SCENE = "cornell_mesh"
W, H, SPP = 1920, 1080, 256
SEED = 1
HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes_per_sample(w, h, spp):
    """fixed record sizes x per-sample counts (tests/golden/work_counters.json,
    produced by tools/make_work_counters.py from the oracle walking the kernel's own structures)."""
    path = os.path.join(ROOT, "tests", "golden", "work_counters.json")
    with open(path) as f:
        c = json.load(f)[SCENE]["per_sample"]
    b = (32.0 * c["sphere_tests"] + 48.0 * c["quad_tests"] + 32.0 * c["node_visits"] + 40.0 * c["tri_tests"]
         + 64.0 * c["shaded_hits"] + 3.0 * c["texel_lookups"])
    return b + 12.0 / spp, c


def measured_traffic_bytes_per_launch():
    """HBM bytes per trace-kernel launch from the rocprofv3 PMC passes, if a summary is committed."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if not os.path.exists(path):
        return None
    with open(path) as f:
        j = json.load(f)
    key = f"{SCENE}_{W}x{H}@{SPP}"
    return j.get(key, {}).get("hbm_bytes_per_launch")


def cpu_baseline(hrt, desc, cam):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    w, h = 480, 270  # bounded sample of the same scene / camera / seed, spp sized for ~15 s of CPU work
    threads = os.cpu_count() or 1
    scene = oracle_lib.OracleScene(desc, oracle_lib.MESH_REF_TREE)
    t0 = time.perf_counter()
    scene.render(cam, w, h, 2, seed=SEED, threads=threads)  # page the scene in, start the thread pool once
    t0 = time.perf_counter()
    scene.render(cam, w, h, 32, seed=SEED, threads=threads)  # calibration
    rate = w * h * 32 / max(time.perf_counter() - t0, 1e-3)
    spp = int(min(1024, max(8, 12.0 * rate / (w * h))))
    t0 = time.perf_counter()
    scene.render(cam, w, h, spp, seed=SEED, threads=threads)
    dt = time.perf_counter() - t0
    return {
        "value": round(w * h * spp / dt / 1e6, 4), "unit": "Msamples/s", "cores": threads, "kind": "port",
        "sample": f"{SCENE} {w}x{h}@{spp} spp, same camera and seed, oracle (reference-shaped KD-tree), "
                  f"{threads} threads over scanlines, {dt:.1f} s",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=SPP, help="samples per pixel per GPU (default: the metric's 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    # One rank per GPU.  HRT_BENCH_SHARE_GPU=1 is a rehearsal mode for a one-GPU box: all ranks share GPU 0 and the
    # collectives run over gloo (staged through host memory), which exercises everything but RCCL itself.
    share = os.environ.get("HRT_BENCH_SHARE_GPU") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world, device_id=device)

    hrt = importlib.import_module("hai719-raytracing_amd")
    hdist = importlib.import_module("hai719-raytracing_amd.dist")
    hrt.init(dev_index)

    spp = args.spp * world  # weak scaling: per-GPU samples stay w*h*args.spp
    host = hrt.HostScene().setup(SCENE, W / H, 1)
    desc = host.flatten()
    cam = hrt.default_camera(W / H)
    scene = hrt.DeviceScene(desc)  # upload: outside the timed region
    stream = torch.cuda.current_stream().cuda_stream

    kernel_ms = []

    def step():
        def fill(buf):
            scene.render_tiles(cam, W, H, spp, SEED, 0, rank, world, buf.data_ptr(), stream)
        frame = hdist.render_frame_distributed(fill, W, H, rank, world, device, on_gpu=True, stream_ptr=stream)
        return frame

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    t0 = time.perf_counter()
    frame = None
    for _ in range(args.steps):
        frame = step()
        kernel_ms.append(scene.last_kernel_ms())  # HIP events on the launch stream (blocks on this step's kernel)
    sync()
    elapsed = time.perf_counter() - t0
    t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share else device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    if rank == 0:
        samples_per_step = W * H * spp
        value = samples_per_step * args.steps / elapsed / 1e6
        bps, counts = algorithmic_bytes_per_sample(W, H, spp)
        launch_samples = W * H * spp / world  # what ONE launch (this rank's tiles) traces
        avg_kernel_s = sum(kernel_ms) / len(kernel_ms) / 1e3
        achieved = bps * launch_samples / avg_kernel_s / 1e9
        traffic = measured_traffic_bytes_per_launch() if (world == 1 and args.spp == SPP) else None
        out = {
            "metric": "Msamples/s (pixels x spp / s), Cornell+mesh 1080p@256spp",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{SCENE} (Cornell box + flamingo_lowpoly.off, KD-tree) {W}x{H} @ {spp} spp, "
                                   f"seed {SEED}, default camera", "spp_per_gpu_share": args.spp,
                       "partition": f"8x8 tiles round-robin over {world} rank(s), one gather to rank 0"},
            "kernel_ms_per_launch": round(avg_kernel_s * 1e3, 3),
            "roofline": {"bound": "hbm", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                         "algorithmic_bytes_per_sample": round(bps, 1),
                         "note": "achieved = SURVEY 8(d) algorithmic bytes / kernel time; those records are served from SGPRs, LDS "
                                 "and L2, so frac can exceed 1; traffic = L2<->fabric bytes per launch from rocprofv3 PMC (path records)"},
        }
        if frame is not None:
            out["frame_mean"] = round(float(frame.mean().item()), 6)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(hrt, desc, cam)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
