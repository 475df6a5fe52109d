"""Build the gfx950 shared library `shasta_amd/csrc/libshasta_hip.so` in-tree with hipcc.

    python -m shasta_amd.build          # (re)build if sources are newer than the .so
    python -m shasta_amd.build --force

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the repo snapshot.
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
LIB = os.path.join(CSRC, "libshasta_hip.so")
SOURCES = ["abi.hip", "bev_gather.hip", "gemm_f32.hip", "gemm_pieces.hip", "anchor.hip", "anchor_mfma.hip", "anchor_split.hip", "pair.hip", "pair_f16.hip", "pair_f16w.hip", "embed_rows.hip", "aff.hip", "aff_pieces.hip", "aff_f16.hip", "forward.hip",
           "voxelize.hip", "shared_conv.hip", "shared_conv_f16.hip", "shared_conv_train.hip", "iou3d.hip", "decode.hip", "train.hip", "pair_bwd.hip", "track.hip", "nms.hip"]
# -ffp-contract=off: every fused multiply-add in the kernels is an explicit fmaf(); products that the
# reference rounds separately stay separately rounded (parity with the PyTorch fp32 forward).
# -fvisibility=hidden: the .so exports exactly the extern "C" functions include/shasta_hip.h declares (its visibility pragma)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# per-file additions.  pair.hip: keep the MFMA accumulators in architectural VGPRs (113 registers instead of 116 + 44 AGPRs:
# 4 waves per SIMD become possible, and a layer's accumulators feed the next layer without v_accvgpr_read)
# gemm_pieces.hip: same switch - with AGPR accumulators the allocator moved all 64 of them through VGPRs in every K slice
EXTRA_FLAGS = {"pair.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "gemm_pieces.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "aff_pieces.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "aff_f16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "pair_f16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"],
               "embed_rows.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
# (shared_conv_train.hip keeps its 144 accumulators per wave in AGPRs: default form)


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _dep_files():
    deps = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".hpp", ".h", ".map"))]
    deps.append(os.path.normpath(os.path.join(os.path.dirname(CSRC), "..", "include", "shasta_hip.h")))
    return [d for d in deps if os.path.exists(d)]


def source_hash():
    """16 hex digits over the names and contents of everything the library is built from; compiled into the library
    (shasta_build_info) so that a stale libshasta_hip.so - sources edited, build failed or forgotten - is refused at load."""
    h = hashlib.sha256()
    for d in _dep_files():
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in _dep_files())


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build")
    os.makedirs(objdir, exist_ok=True)
    srchash = source_hash()

    def cc(src):
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        extra = ['-DSHASTA_SOURCE_HASH="%s"' % srchash] if src == "abi.hip" else []
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + extra + ["-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(cc, SOURCES))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,--version-script=" + os.path.join(CSRC, "exports.map"), "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
