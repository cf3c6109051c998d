"""Where the trace kernels spill: per kernel, the compiler's resource summary and every scratch / lane-spill
instruction attributed to its source line (hipcc -gline-tables-only -S).  Runs in the build container (no GPU):

    python tools/spill_map.py > profiles/r02_spill_map.txt
"""
import collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "hai719-raytracing_amd", "csrc", "hrt_api.hip")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "--offload-arch=gfx950", "-Wno-unused-function", "-Wno-bitwise-instead-of-logical",
         "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]   # the device flags of hai719-raytracing_amd/Makefile
KERNELS = ["hrt_wgstream_kernel", "hrt_wgstream_kernel_lights", "hrt_wgstream_kernel_sph", "hrt_wgstream_kernel_lights_sph", "hrt_trace_kernel", "hrt_trace_kernel_lights", "hrt_trace2_kernel", "hrt_trace2_kernel_lights"]
tmp = tempfile.mkdtemp()
asm = os.path.join(tmp, "hrt.s")
r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-Rpass-analysis=kernel-resource-usage", "-gline-tables-only", "-S", "--cuda-device-only", "-o", asm, SRC],
                   capture_output=True, text=True)
usage = collections.defaultdict(dict)
cur = None
for line in r.stderr.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m: cur = m.group(1)
    m = re.search(r"remark:\s+(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur: usage[cur][m.group(1)] = int(m.group(2))
s = open(asm).read()
files = {int(m.group(1)): (m.group(3) or m.group(2)).split("/")[-1] for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s)}
srcs = {}
def src_line(f, n):
    p = os.path.join(ROOT, "hai719-raytracing_amd", "csrc", f)
    if f not in srcs: srcs[f] = open(p).read().split("\n") if os.path.exists(p) else None
    return srcs[f][n - 1].strip()[:110] if srcs[f] and 0 < n <= len(srcs[f]) else ""
print("# Spill map of the trace kernels (tools/spill_map.py; hipcc " + " ".join(FLAGS) + ")")
print("# scratch_* = VGPR spills to the private segment; v_readlane/v_writelane = SGPR spills kept in VGPR lanes (no memory traffic).")
for k in KERNELS:
    start = s.find("\n" + k + ":")
    if start < 0: continue
    body = s[start:s.index(".Lfunc_end", start)].split("\n")
    print(f"\n## {k}: " + ", ".join(f"{a} {b}" for a, b in usage[k].items()))
    loc, cnt = None, collections.Counter()
    for l in body:
        t = l.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
        if m: loc = (files.get(int(m.group(1)), "?"), int(m.group(2))); continue
        op = t.split()[0] if t else ""
        if op.startswith("scratch_"): cnt[(loc, "scratch_load" if "load" in op else "scratch_store")] += 1
    tot = collections.Counter()
    for (loc, op), n in cnt.items(): tot[op] += n
    print(f"   scratch instructions: {dict(tot)}")
    for (loc, op), n in sorted(cnt.items(), key=lambda x: (x[0][0] or ("", 0), x[0][1])):
        f, ln = loc if loc else ("?", 0)
        print(f"   {op:13s} x{n:3d}  {f}:{ln}  {src_line(f, ln)}")
