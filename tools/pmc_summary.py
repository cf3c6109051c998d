"""rocprofv3 outputs of tools/profile.sh -> the summaries committed under profiles/ (run on the GPU box by profile.sh):

    python tools/pmc_summary.py <dir of one configuration> <tag> <cfg> <scene> <spp> <w> <h>

writes <dir>/../<tag>_<cfg>_pmc.json (counters of the trace kernel for ONE launch + derived figures + the hash of the kernel
sources they were measured on: bench.py uses them only for a build with the same hash) and <tag>_<cfg>_kernel_stats.csv (the
--kernel-trace --stats table).  Units, per /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KiB at the
L2's fabric side (Infinity-Cache hits included); SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / SQ_WAIT_* count quad-cycles summed over
all SIMDs; SQ_BUSY_CYCLES counts cycles summed over the 32 shader engines.

Fabric bytes: the guide's factor of 2 on FETCH_SIZE is for wide coalesced streaming reads and says other patterns are
uncalibrated.  This kernel's traffic is a gather of 16-byte groups from 128-byte path records, so the bytes are taken from the
request-size counters instead -- 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B, and 64 x WRREQ_64B + 32 x the other write
requests -- which tools/calibrate_traffic.sh checks against a known byte count on exactly that pattern
(profiles/<tag>_traffic_calibration.json; `fetch_factor` below is the resulting ratio to FETCH_SIZE, reported, not assumed)."""
import collections, csv, glob, hashlib, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def source_sha16():
    """Hash of everything libhrt.so is compiled from (the same function is in bench.py)."""
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "hai719-raytracing_amd", "csrc", "*"))) + [os.path.join(ROOT, "include", "hrt.h"),
                                                                                          os.path.join(ROOT, "hai719-raytracing_amd", "Makefile")]
    for f in files:
        h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    cfg_dir, tag, cfg, scene, spp, w, h = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    out_dir = os.path.dirname(os.path.abspath(cfg_dir))
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(int))
    for f in glob.glob(os.path.join(cfg_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k][r["Counter_Name"]] += 1
    trace = [k for k in acc if k.startswith("hrt_wgstream") or k.startswith("hrt_trace")]
    if not trace:
        raise SystemExit("no trace kernel in the PMC output")
    kernel = max(trace, key=lambda k: acc[k].get("SQ_WAVE_CYCLES", 0.0))
    c = {name: v / max(1, launches[kernel][name]) for name, v in acc[kernel].items()}   # per launch
    samples = w * h * spp
    d = {}
    if "TCC_EA0_RDREQ_sum" in c and "TCC_EA0_WRREQ_sum" in c:
        n32, n64, n128 = c.get("TCC_EA0_RDREQ_32B_sum", 0.0), c.get("TCC_EA0_RDREQ_64B_sum", 0.0), c.get("TCC_EA0_RDREQ_128B_sum", 0.0)
        other = max(c["TCC_EA0_RDREQ_sum"] - n32 - n64 - n128, 0.0)
        d["fabric_read_bytes_per_launch"] = 32.0 * n32 + 64.0 * n64 + 128.0 * n128 + 64.0 * other
        w64 = c.get("TCC_EA0_WRREQ_64B_sum", 0.0)
        d["fabric_write_bytes_per_launch"] = 64.0 * w64 + 32.0 * max(c["TCC_EA0_WRREQ_sum"] - w64, 0.0)
        d["fabric_bytes_per_launch"] = d["fabric_read_bytes_per_launch"] + d["fabric_write_bytes_per_launch"]
        d["fabric_bytes_per_sample"] = round(d["fabric_bytes_per_launch"] / samples, 1)
        if "FETCH_SIZE" in c: d["fetch_factor"] = round(d["fabric_read_bytes_per_launch"] / (c["FETCH_SIZE"] * 1024.0), 3)
        if "WRITE_SIZE" in c: d["write_factor"] = round(d["fabric_write_bytes_per_launch"] / (c["WRITE_SIZE"] * 1024.0), 3)
    if "SQ_ACTIVE_INST_VALU" in c and "SQ_BUSY_CYCLES" in c:
        d["valu_active_simd_cycles"] = 4.0 * c["SQ_ACTIVE_INST_VALU"]            # quad-cycles -> cycles, summed over SIMDs
        d["simd_cycles"] = c["SQ_BUSY_CYCLES"] / 32.0 * 1024.0                  # busy cycles per shader engine x 1024 SIMDs
        d["valu_busy_frac"] = round(d["valu_active_simd_cycles"] / d["simd_cycles"], 4)
    if "SQ_THREAD_CYCLES_VALU" in c and "SQ_ACTIVE_INST_VALU" in c:
        d["valu_lane_utilisation"] = round(c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"]), 4)
    if "SQ_INSTS_VALU" in c:
        d["valu_wave_insts_per_sample"] = round(c["SQ_INSTS_VALU"] / samples, 2)
    if "SQ_INSTS_SALU" in c:
        d["salu_wave_insts_per_sample"] = round(c["SQ_INSTS_SALU"] / samples, 2)
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
        d["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        d["wait_any_share_of_wave_cycles"] = round(c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"], 4)
    stats = glob.glob(os.path.join(cfg_dir, "trace", "**", "*kernel_stats.csv"), recursive=True)
    avg_ms = None
    if stats:
        shutil.copy(stats[0], os.path.join(out_dir, f"{tag}_{cfg}_kernel_stats.csv"))
        for r in csv.DictReader(open(stats[0])):
            if r["Name"] == kernel:
                avg_ms = float(r["AverageNs"]) / 1e6
                d["kernel_trace_avg_ms"] = round(avg_ms, 3)
                d["kernel_trace_calls"] = int(r["Calls"])
    # Vector issue rate against a MEASURED peak.  valu_busy_frac above prices every vector instruction at 4 cycles of its SIMD (64 lanes
    # on 16) -- on MI355X a simple fp32 instruction takes 2.7 (tools/micro/valu_rate.hip: 1.30 ns per SIMD with 4 waves per SIMD, the
    # streaming kernel's occupancy), so that figure is NOT bounded by 1 (random_spheres: 1.17).  This one is: wave instructions per
    # second over 1024 SIMDs x the measured rate of the cheapest instruction (half-rate and transcendental ones cost 1.4-2.6 x that).
    cal = os.path.join(ROOT, "profiles", f"{tag}_valu_rate.json")
    if avg_ms and "SQ_INSTS_VALU" in c and os.path.exists(cal):
        peak = json.load(open(cal))["peak_wave_insts_per_ns"]
        d["valu_insts_per_ns"] = round(c["SQ_INSTS_VALU"] / (avg_ms * 1e6), 1)
        d["valu_peak_insts_per_ns"] = peak
        d["valu_issue_frac"] = round(d["valu_insts_per_ns"] / peak, 4)
    if avg_ms and "fabric_bytes_per_launch" in d:
        d["fabric_gbps_at_trace_time"] = round(d["fabric_bytes_per_launch"] / (avg_ms / 1e3) / 1e9, 1)
        d["msamples_per_s_at_trace_time"] = round(samples / avg_ms / 1e3, 1)
    j = {"kernel": kernel, "cfg": cfg, "scene": scene, "w": w, "h": h, "spp": spp, "source_sha16": source_sha16(),
         "config": f"{scene} {w}x{h}@{spp}, ONE launch per PMC pass (tools/prof_one.py under rocprofv3 --pmc, each counter group in its own run; tools/profile.sh {tag})",
         "source": f"profiles/{tag}_{cfg}_pmc.json (tools/profile.sh {tag} + tools/pmc_summary.py; kernel time: profiles/{tag}_{cfg}_kernel_stats.csv)",
         "counters_per_launch": c, "derived": d}
    json.dump(j, open(os.path.join(out_dir, f"{tag}_{cfg}_pmc.json"), "w"), indent=1)
    print(cfg, json.dumps(d))
