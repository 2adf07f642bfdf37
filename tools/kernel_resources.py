"""tools/kernel_resources.py [extra hipcc flags]: registers / LDS / scratch of every kernel of csrc/fisher_rast.hip (from the
gfx950 assembly's metadata; no GPU needed).  Waves per SIMD allowed by registers = min(8, 512 // alloc), alloc = ceil8(vgpr + agpr)."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "fisher-nerf-customized_amd", "csrc", sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".hip") else "fisher_rast.hip")
extra = [a for a in sys.argv[1:] if not a.endswith(".hip")]
out = "/tmp/kernel_resources.s"
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off",
                       "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-slp-vectorize", *extra, "-S", "--cuda-device-only", "-o", out, src],
                      stderr=subprocess.DEVNULL)
s = open(out).read()
pat = re.compile(r"\.agpr_count:\s+(\d+).*?\.group_segment_fixed_size:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+)"
                 r".*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", re.S)
rows = []
for ag, lds, name, priv, sg, vg in pat.findall(s):
    try:
        dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except FileNotFoundError:
        dn = name
    dn = re.sub(r"\(.*", "", dn).replace("void ", "")
    alloc = (int(vg) + 7) // 8 * 8          # .vgpr_count already includes the AGPRs
    rows.append((dn, int(vg), int(ag), int(sg), int(lds), int(priv), min(8, 512 // max(alloc, 8))))
for r in sorted(rows):
    print(f"{r[0][:64]:64s} vgpr {r[1]:4d} (agpr {r[2]:3d}) sgpr {r[3]:3d} lds {r[4]:6d} scratch {r[5]:4d} waves/SIMD {r[6]}")
