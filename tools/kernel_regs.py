"""Print registers / scratch / LDS of every kernel in a .hip file (developer tool): python tools/kernel_regs.py file.hip [extra flags]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffast-math", "-fno-finite-math-only", "-I" + os.path.join(root, "include")] + sys.argv[2:]
asm = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-S", "--cuda-device-only", src, "-o", "-"], capture_output=True, text=True).stdout
cur = {}
for line in asm.splitlines():
    m = re.match(r"\s+\.(name|vgpr_count|agpr_count|private_segment_fixed_size|group_segment_fixed_size|sgpr_count):\s+(\S+)", line)
    if m:
        if m.group(1) == "name" and "name" in cur and "vgpr_count" in cur:
            print("%-70s vgpr %4s agpr %4s scratch %5s lds %6s" % (cur["name"][:70], cur.get("vgpr_count"), cur.get("agpr_count"), cur.get("private_segment_fixed_size"), cur.get("group_segment_fixed_size")))
            cur = {}
        cur[m.group(1)] = m.group(2)
if "vgpr_count" in cur:
    print("%-70s vgpr %4s agpr %4s scratch %5s lds %6s" % (cur["name"][:70], cur.get("vgpr_count"), cur.get("agpr_count"), cur.get("private_segment_fixed_size"), cur.get("group_segment_fixed_size")))
