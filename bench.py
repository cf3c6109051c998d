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

Rank 0 prints ONE JSON line.  What its evidence keys mean:
  roofline      memory side of the dominant kernel.  `traffic` = bytes per launch that crossed the L2 <-> fabric boundary, from
                the committed rocprofv3 PMC passes (profiles/r03_<cfg>_pmc.json; separate --pmc runs): read and write requests
                by size (TCC_EA0_RDREQ_32B / _64B / _128B, WRREQ_64B / others), checked against a known byte count on the path
                pool's own access pattern (profiles/r03_traffic_calibration.json).  The summary carries the hash of the kernel
                sources it was measured on: a build with another hash gets traffic = null (never a stale number).
                Infinity-Cache hits are in that count, so it is an UPPER bound on HBM bytes.  achieved = traffic / the
                kernel's launch duration measured live here with HIP events on the launch stream; frac = achieved / 8 TB/s.  `algorithmic` keeps SURVEY 8(d)'s convention (fixed record sizes x per-sample work counters)
                as information only: those records are served from SGPRs, LDS and L2, not from HBM.
  valu          the compute side: vector wave instructions per second (SQ_INSTS_VALU of the committed PMC run / the live kernel
                time) over the MEASURED issue peak of the part (tools/micro/valu_rate.hip: 1024 SIMDs x one simple instruction
                per 1.30 ns at this occupancy), with the lane utilisation of that work beside it (same PMC runs).
  value_host_to_host   the same frame through hrt_render: gamma, tile assemble and the D2H copy of the frame included.
  without_pruning      the kernel rate of the same frame on a scene created with HRT_PRUNE=0 (no exact path pruning), and the
                       check that the two frames are bit-identical: `value` includes an optimisation that skips provably
                       irrelevant work, this is the rate without it.
  other_configs        kernel time of BASELINE's other configurations at their true sizes (one launch each).
  cpu_baseline  the CPU oracle (a port of the reference algorithm, reference-shaped KD-tree) timed on this host on a
                bounded sample of the same workload, threaded three ways -- a reported baseline, not the target.
"""
import argparse
import glob
import hashlib
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


PROFILE_TAG = "r03"


def source_sha16():
    """Hash of everything libhrt.so is compiled from (the same function is in tools/pmc_summary.py)."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "hai719-raytracing_amd", "csrc", "*"))) + [os.path.join(ROOT, "include", "hrt.h"),
                                                                                          os.path.join(ROOT, "hai719-raytracing_amd", "Makefile")]
    for f in files:
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def committed_pmc(cfg, scene, w, h, spp):
    """(summary, note): counters of the trace kernel for this configuration from the committed rocprofv3 PMC summary
    (profiles/r03_<cfg>_pmc.json, written by tools/profile.sh + tools/pmc_summary.py) -- only when it was measured on the
    sources this build is compiled from; otherwise (None, why)."""
    path = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_{cfg}_pmc.json")
    if not os.path.exists(path):
        return None, f"no profiles/{PROFILE_TAG}_{cfg}_pmc.json"
    with open(path) as f:
        j = json.load(f)
    if (j.get("scene"), j.get("w"), j.get("h"), j.get("spp")) != (scene, w, h, spp):
        return None, "the committed profile is of another workload"
    sha = source_sha16()
    if j.get("source_sha16") != sha:
        return None, f"the committed profile was measured on kernel sources {j.get('source_sha16')}, this build is {sha}: re-run tools/profile.sh"
    return j, None


def valu_of(pmc, kernel_ms):
    """Vector-instruction issue of the trace kernel against a MEASURED peak: wave instructions of one launch (SQ_INSTS_VALU, committed
    PMC run) / this run's live kernel time, over 1024 SIMDs x the rate of the cheapest vector instruction at this occupancy
    (tools/micro/valu_rate.hip -> profiles/r03_valu_rate.json: 1.30 ns per instruction per SIMD).  Round 2's figure -- SQ_ACTIVE_INST_VALU
    x 4 cycles over the SIMD cycles -- is kept as `frac_at_4_cycles_per_inst`; it is not bounded by 1 on this part (a simple fp32
    instruction takes 2.7 cycles, not 4)."""
    d, c = pmc["derived"], pmc["counters_per_launch"]
    out = {"lane_utilisation": d["valu_lane_utilisation"], "valu_wave_insts_per_sample": d["valu_wave_insts_per_sample"],
           "frac_at_4_cycles_per_inst": d["valu_busy_frac"]}
    cal = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_valu_rate.json")
    if os.path.exists(cal) and "SQ_INSTS_VALU" in c:
        with open(cal) as f:
            peak = json.load(f)["peak_wave_insts_per_ns"]
        out["insts_per_ns"] = round(c["SQ_INSTS_VALU"] / (kernel_ms * 1e6), 1)
        out["peak_insts_per_ns"] = peak
        out["frac"] = round(out["insts_per_ns"] / peak, 4)
        out["note"] = ("wave instructions per launch (SQ_INSTS_VALU, committed PMC run of the same sources) / live kernel time, over the measured issue "
                       "rate of the cheapest vector instruction on 1024 SIMDs at 4 waves per SIMD (profiles/r03_valu_rate.json); half-rate and "
                       "transcendental instructions cost 1.4-2.6 x that, so this is a LOWER bound on how busy the vector units are")
    else:
        out["frac"] = None
        out["note"] = "no profiles/r03_valu_rate.json: the issue peak is not calibrated"
    return out


def cpu_baseline(hrt, desc, cam):
    """The oracle on a bounded sample, ~8 s per leg: (1) a pool of hardware_concurrency threads over scanlines with
    per-path random streams -- the `value`; (2) one std::thread per scanline, all spawned at once, as the reference does
    (main.cpp:232-238); (3) the pool drawing every random number from ONE shared generator, the reference's
    random_float() (Functions.cpp:4-8; mutex-protected here, a data race there).  The sample is the WHOLE 1920 x 1080 frame at
    fewer samples per pixel: the work items are scanlines (as in the reference), and 1080 of them over the host's threads
    leaves no leg bound by its tail (a 270-line sample gave 256 threads one line each and 14 of them two)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib

    w, h = W, H  # the frame of the metric, same scene / camera / seed; bounded through the samples per pixel
    threads = os.cpu_count() or 1
    scene = oracle_lib.OracleScene(desc, oracle_lib.MESH_REF_TREE)
    scene.render(cam, w, h, 1, seed=SEED, threads=threads)  # page the scene in, start the thread pool once
    t0 = time.perf_counter()
    scene.render(cam, w, h, 1, seed=SEED, threads=threads)  # calibration
    rate = w * h / max(time.perf_counter() - t0, 1e-3)
    spp = int(min(256, max(1, 8.0 * rate / (w * h))))

    def leg(nthreads, flags, spp_):
        t0 = time.perf_counter()
        scene.render(cam, w, h, spp_, seed=SEED, threads=nthreads, flags=flags)
        dt = time.perf_counter() - t0
        return round(w * h * spp_ / dt / 1e6, 4), round(dt, 1)

    pool, dt_pool = leg(threads, 0, spp)
    per_line, dt_line = leg(-1, 0, spp)
    shared_threads = min(threads, 8)  # the reference was written for a desktop CPU; with hundreds of threads a lock only measures the lock
    t0 = time.perf_counter()
    scene.render(cam, w, h, 1, seed=SEED, threads=shared_threads, flags=1 << 16)
    shared_rate = w * h / max(time.perf_counter() - t0, 1e-3)
    shared_spp = int(min(spp, max(1, 6.0 * shared_rate / (w * h))))
    shared, dt_shared = leg(shared_threads, 1 << 16, shared_spp)
    return {
        "value": pool, "unit": "Msamples/s", "cores": threads, "kind": "port",
        "sample": f"{SCENE} {w}x{h}@{spp} spp, same camera and seed, oracle (reference-shaped KD-tree), "
                  f"{h} scanline work items over {threads} threads ({h / threads:.1f} per thread), {dt_pool} s",
        "thread_per_scanline": {"value": per_line, "threads": h, "seconds": dt_line,
                                "note": "one std::thread per scanline, all started at once (main.cpp:232-238)"},
        "shared_rng": {"value": shared, "threads": shared_threads, "spp": shared_spp, "seconds": dt_shared,
                       "note": "every draw from one process-wide mt19937 (random_float(), Functions.cpp:4-8; mutex here, a race there)"},
    }


OTHER_CONFIGS = [("cfg1", "cornell_box", 256, 256, 4), ("cfg2", "cornell_mesh", 1920, 1080, 64), ("cfg3", "random_spheres", 1920, 1080, 256),
                 ("cfg4", "mesh_in_box", 3840, 2160, 512), ("cfg5", "backrooms_pool", 3840, 2160, 1024)]


def frame_mean(img):
    """Mean of a frame in float64 over the float32 pixels: pixels are deterministic (keyed random streams, ordered sums), so this
    number is too, and tests/golden/bench_frame_means.json holds it for every configuration bench.py times."""
    import numpy as np
    return float(np.asarray(img, dtype=np.float64).mean())


def committed_frame_means():
    path = os.path.join(ROOT, "tests", "golden", "bench_frame_means.json")
    if not os.path.exists(path):
        return {}
    with open(path) as f:
        return json.load(f)


def other_configs(hrt):
    """Kernel time of BASELINE.json's other configurations at their true sizes on this GPU: one launch each (~12 s).  What is
    timed is also CHECKED: every frame must be finite and its mean must equal the committed one exactly (tolerance 0)."""
    import numpy as np
    out = {}
    means = committed_frame_means()
    for tag, name, w, h, spp in OTHER_CONFIGS:
        dev = hrt.DeviceScene(hrt.HostScene().setup(name, w / h, 1).flatten())
        cam = hrt.default_camera(w / h)
        dev.render(cam, 64, 64, 1, SEED)  # first-launch costs out of the way
        img, st = dev.render(cam, w, h, spp, SEED)
        if not np.isfinite(img).all():
            raise SystemExit(f"bench.py: {tag} ({name} {w}x{h}@{spp}) rendered non-finite pixels")
        mean = frame_mean(img)
        key = f"{name} {w}x{h}@{spp} seed {SEED}"
        if key in means and means[key] != mean:
            raise SystemExit(f"bench.py: {tag}: frame mean {mean!r} differs from the committed {means[key]!r} ({key})")
        out[tag] = {"scene": name, "size": f"{w}x{h}@{spp}", "kernel_ms": round(st.kernel_ms, 2),
                    "msamples_per_s": round(w * h * spp / st.kernel_ms / 1e3, 1), "frame_mean": mean,
                    "frame_mean_checked": key in means}
        pmc, why = committed_pmc(tag, name, w, h, spp)
        if pmc:
            d = pmc["derived"]
            out[tag]["valu"] = {k: valu_of(pmc, st.kernel_ms)[k] for k in ("frac", "lane_utilisation", "valu_wave_insts_per_sample")}
            if "fabric_bytes_per_launch" in d:
                out[tag]["fabric_frac_of_hbm_peak"] = round(d["fabric_bytes_per_launch"] / (st.kernel_ms / 1e3) / 1e9 / HBM_PEAK_GBS, 4)
            out[tag]["kernel"] = pmc["kernel"]
        else:
            out[tag]["profile"] = why
        dev.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=SPP, help="samples per pixel per GPU (default: the metric's 256)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip value_host_to_host and other_configs (profiling runs)")
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
        pmc, pmc_note = committed_pmc("cfg2_256", SCENE, W, H, SPP) if (world == 1 and args.spp == SPP) else (None, "profiles are of the 1-GPU metric line")
        traffic = pmc["derived"].get("fabric_bytes_per_launch") if pmc else None
        achieved = traffic / avg_kernel_s / 1e9 if traffic else None
        roofline = {
            "bound": "hbm", "achieved": None if achieved is None else round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
            "traffic_source": pmc_note if not pmc else pmc["source"],
            "traffic_calibration": None if not pmc else {"fetch_factor": pmc["derived"].get("fetch_factor"), "write_factor": pmc["derived"].get("write_factor"),
                                                       "see": f"profiles/{PROFILE_TAG}_traffic_calibration.json"},
            "algorithmic": {"bytes_per_sample": round(bps, 1), "bytes_per_launch": round(bps * launch_samples),
                            "gbps": round(bps * launch_samples / avg_kernel_s / 1e9, 1),
                            "note": "SURVEY 8(d) convention (record sizes x work counters): served from SGPRs / LDS / L2, not a bandwidth"},
            "note": "achieved = L2<->fabric bytes per launch (requests by size, Infinity-Cache hits included: an upper "
                    "bound on HBM bytes) / the kernel's live HIP-event time; see attainable and valu",
        }
        coop = os.path.join(ROOT, "profiles", f"{PROFILE_TAG}_record_coop.json")
        if os.path.exists(coop):  # what a kernel that does NOTHING but hit visits of this record pool reaches on the same counters
            with open(coop) as f:
                cj = json.load(f)["counters"]
            roofline["attainable"] = {"gbps": round(cj["gather"]["fabric_tbps"] * 1e3, 1), "gbps_cooperative_access": round(cj["coop"]["fabric_tbps"] * 1e3, 1),
                                      "frac_of_attainable": None if achieved is None else round(achieved / (cj["gather"]["fabric_tbps"] * 1e3), 3),
                                      "note": "tools/micro/record_coop.hip under the same request counters (profiles/r03_record_coop.json): 5 group loads + 5 group stores per "
                                              "lane on random 128-byte records of a 128 MiB pool, nothing else -- the fabric rate this access pattern reaches on this part"}
        valu = valu_of(pmc, avg_kernel_s * 1e3) if pmc else None
        out = {
            "metric": "Msamples/s (pixels x spp / s), Cornell+mesh 1080p@256spp",
            "value": round(value, 2), "unit": "Msamples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{SCENE} (Cornell box + flamingo_lowpoly.off, KD-tree) {W}x{H} @ {spp} spp, "
                                   f"seed {SEED}, default camera", "spp_per_gpu_share": args.spp,
                       "partition": f"8x8 tiles round-robin over {world} rank(s), one gather to rank 0"},
            "kernel_ms_per_launch": round(avg_kernel_s * 1e3, 3),
            "roofline": roofline, "valu": valu,
        }
        out["source_sha16"] = source_sha16()
        if frame is not None:
            import numpy as np
            host_frame = frame.cpu().numpy()
            if not np.isfinite(host_frame).all():
                raise SystemExit("bench.py: the timed frame has non-finite pixels")
            out["frame_mean"] = frame_mean(host_frame)
            key = f"{SCENE} {W}x{H}@{spp} seed {SEED}"
            means = committed_frame_means()
            if key in means and means[key] != out["frame_mean"]:
                raise SystemExit(f"bench.py: frame mean {out['frame_mean']!r} differs from the committed {means[key]!r} ({key})")
            out["frame_mean_checked"] = key in means
        if world == 1 and not args.no_extras:
            import numpy as np
            host_img, st = scene.render(cam, W, H, spp, SEED, flags=hrt.FLAG_GAMMA)  # once to size the library's buffers
            t1 = time.perf_counter()
            host_img, st = scene.render(cam, W, H, spp, SEED, flags=hrt.FLAG_GAMMA)
            dt = time.perf_counter() - t1
            out["value_host_to_host"] = {"value": round(W * H * spp / dt / 1e6, 2), "unit": "Msamples/s", "ms": round(dt * 1e3, 2),
                                         "note": "hrt_render: kernel + gamma + tile assemble + D2H of the 24.9 MB frame, host call to host buffer"}
            assert np.isfinite(host_img).all()
            # The exact path pruning of round 3 (a path ends when its throughput is exactly zero; the last segment of a
            # path in an unlit scene is followed only if its closest sphere / square hit emits) removes work without changing a bit.
            # Reported beside the metric so that nobody has to take that on trust: the same kernels on a scene created with the
            # pruning switched off (HRT_PRUNE=0), the frame compared bit for bit.
            os.environ["HRT_PRUNE"] = "0"
            try:
                plain = hrt.DeviceScene(desc)
            finally:
                del os.environ["HRT_PRUNE"]
            plain.render(cam, 64, 64, 1, SEED)
            plain_img, plain_st = plain.render(cam, W, H, spp, SEED, flags=hrt.FLAG_GAMMA)
            if not np.array_equal(plain_img, host_img):
                raise SystemExit("bench.py: the frame rendered without the pruning differs from the frame rendered with it")
            out["without_pruning"] = {"value": round(W * H * spp / plain_st.kernel_ms / 1e3, 2), "unit": "Msamples/s (kernel time)", "kernel_ms": round(plain_st.kernel_ms, 3),
                                      "frame": "bit-identical to the pruned frame (checked in this run)"}
            plain.close()
            out["other_configs"] = other_configs(hrt)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(hrt, desc, cam)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
