"""Build libsed_hip.so (hipcc, gfx950) in-tree.  `python -m transformer4sed_amd.build [--force]`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libsed_hip.so")
SOURCES = ["gemm.hip", "attention.hip", "relpos_attention.hip", "norm_elem.hip", "frontend.hip", "pmam.hip", "dasm.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# -ffast-math (re-association, approximate division / sqrt) only where MFMA operand rounding dominates the error anyway: the GEMM
# epilogues, the attention softmax and the PMAM branch (16-bit NHWC activations).  The fp32 LayerNorm / pooling / loss / optimizer
# kernels (norm_elem.hip) and the frontend are built with IEEE semantics (NaN / inf are meaningful on those paths: a fully padded
# clip yields a NaN weak output like the reference).
FAST = ["-ffast-math", "-fno-finite-math-only"]
# attention kernels: MFMA accumulators stay in (unified-file) VGPRs -- the AGPR form costs a v_accvgpr_read/write per softmax
# operand.  The GEMM file is left to the compiler: its 128x128-per-wave kernel needs the AGPR half for its 256 accumulators.
FILE_FLAGS = {"pmam.hip": FAST,
              "gemm.hip": FAST, "attention.hip": FAST + ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
              "relpos_attention.hip": FAST + ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "sed_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    # headers every source includes: a newer header rebuilds everything, otherwise only the sources newer than their object
    hdrs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "sed_hip.h"),
                                                                                     os.path.abspath(__file__)]
    hdr_t = max(os.path.getmtime(h) for h in hdrs)
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        objs.append(obj)
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(hdr_t, os.path.getmtime(os.path.join(CSRC, src))):
            continue
        cmd = [_hipcc(), *FLAGS, *FILE_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out.strip():
            print(out.decode())
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", LIB]
    subprocess.check_call(cmd)
    if verbose:
        print(f"built {LIB}")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
