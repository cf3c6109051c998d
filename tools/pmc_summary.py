"""rocprofv3 outputs of tools/profile.sh -> the summaries committed under profiles/ (run on the GPU box, after profile.sh):

    python tools/pmc_summary.py gpurun_out/prof_<tag> <tag>

writes gpurun_out/prof_<tag>/<tag>_pmc.json (counters of the dominant trace kernel per launch + derived figures, read by
bench.py as profiles/r02_pmc.json) and <tag>_kernel_stats.csv (the --kernel-trace --stats table).  Units, per
/opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KiB at the L2's fabric side (Infinity-Cache hits
included); FETCH_SIZE counts 128-B requests at 64 B on gfx950, so it is doubled; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES /
SQ_WAIT_* count quad-cycles summed over all SIMDs; SQ_BUSY_CYCLES counts cycles summed over the 32 shader engines."""
import collections, csv, glob, json, os, shutil, sys

out_dir, tag = sys.argv[1], sys.argv[2]
SCENE, W, H, SPP = "cornell_mesh", 1920, 1080, 256
acc = collections.defaultdict(lambda: collections.defaultdict(float))
launches = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(os.path.join(out_dir, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        launches[k][r["Counter_Name"]] += 1
trace = [k for k in acc if k.startswith("hrt_wgstream") or k.startswith("hrt_trace")]
if not trace:
    raise SystemExit("no trace kernel in the PMC output")
kernel = max(trace, key=lambda k: acc[k].get("SQ_WAVE_CYCLES", 0.0))
c = {name: v / max(1, launches[kernel][name]) for name, v in acc[kernel].items()}   # per launch
samples = W * H * SPP
d = {}
if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
    d["fabric_bytes_per_launch"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
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
stats = glob.glob(os.path.join(out_dir, "trace", "**", "*kernel_stats.csv"), recursive=True)
avg_ms = None
if stats:
    shutil.copy(stats[0], os.path.join(out_dir, f"{tag}_kernel_stats.csv"))
    for r in csv.DictReader(open(stats[0])):
        if r["Name"] == kernel:
            avg_ms = float(r["AverageNs"]) / 1e6
            d["kernel_trace_avg_ms"] = round(avg_ms, 3)
            d["kernel_trace_calls"] = int(r["Calls"])
if avg_ms and "fabric_bytes_per_launch" in d:
    d["fabric_gbps_at_trace_time"] = round(d["fabric_bytes_per_launch"] / (avg_ms / 1e3) / 1e9, 1)
j = {"kernel": kernel, "config": f"{SCENE} {W}x{H}@{SPP}, ONE launch per PMC pass (bench.py --steps 1 --warmup 0 --no-extras under rocprofv3 --pmc, "
                                 f"each counter group in its own run; tools/profile.sh {tag})",
     "source": f"profiles/{tag}_pmc.json (tools/profile.sh {tag} + tools/pmc_summary.py; kernel time: profiles/{tag}_kernel_stats.csv)",
     "counters_per_launch": c, "derived": d}
json.dump(j, open(os.path.join(out_dir, f"{tag}_pmc.json"), "w"), indent=1)
print(json.dumps(d, indent=1))
