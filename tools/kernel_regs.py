"""Print registers / scratch / LDS of every kernel in a .hip file (developer tool): python tools/kernel_regs.py file.hip [extra flags]"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
sys.path.insert(0, root)
from transformer4sed_amd.build import FLAGS, FILE_FLAGS   # the flags the library itself is built with (per file)
flags = [f for f in FLAGS if f != "-fPIC"] + FILE_FLAGS.get(os.path.basename(src), []) + ["-I" + os.path.join(root, "include")] + sys.argv[2:]
asm = subprocess.run(["/opt/rocm/bin/hipcc", *flags, "-S", "--cuda-device-only", src, "-o", "-"], capture_output=True, text=True).stdout
cur = {}


def flush():
    if "name" in cur and "vgpr_count" in cur:
        print("%-70s vgpr %4s scratch %5s lds %6s" % (cur["name"][:70], cur.get("vgpr_count"), cur.get("private_segment_fixed_size"),
                                                      cur.get("group_segment_fixed_size")))
    cur.clear()


in_meta = False
for line in asm.splitlines():
    if line.startswith("amdhsa.kernels:"):
        in_meta = True
    if not in_meta:
        continue
    if re.match(r"\s+- \.", line):
        flush()
    m = re.match(r"\s+(?:- )?\.(name|vgpr_count|private_segment_fixed_size|group_segment_fixed_size):\s+(\S+)", line)
    if m:
        cur[m.group(1)] = m.group(2)
flush()
