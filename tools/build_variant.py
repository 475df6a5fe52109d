"""Build a variant of the library for A/B runs: one source recompiled with extra -D flags, linked with the objects of the
normal build.  usage: python tools/build_variant.py <name> <source.hip> [-DFLAG ...]  ->  tools/probes/_bin/libshasta_<name>.so
(load it with SHASTA_HIP_LIB=<path>, see tools/gpu_ab.sh).  --export-all links without the export map (diagnostic builds that add an
extern "C" accessor of their own, e.g. -DPAIR_STAMP)."""
import os
import subprocess
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from shasta_amd import build as B  # noqa: E402


def main():
    name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
    export_all = "--export-all" in flags
    flags = [f for f in flags if f != "--export-all"]
    B.build()
    objdir = os.path.join(B.CSRC, "build")
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "probes", "_bin")
    os.makedirs(out, exist_ok=True)
    variant = {}
    for one in src.split(","):  # several sources (comma-separated) take the same flags
        obj = os.path.join(out, "%s_%s.o" % (name, one.replace(".hip", "")))
        cmd = [B._hipcc()] + B.FLAGS + B.EXTRA_FLAGS.get(one, []) + flags + ["-c", os.path.join(B.CSRC, one), "-o", obj]
        subprocess.run(cmd, check=True)
        variant[one] = obj
    objs = [variant.get(s, os.path.join(objdir, s.replace(".hip", ".o"))) for s in B.SOURCES]
    lib = os.path.join(out, "libshasta_%s.so" % name)
    vs = [] if export_all else ["-Wl,--version-script=" + os.path.join(B.CSRC, "exports.map")]
    subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC"] + vs + ["-o", lib] + objs, check=True)
    print(lib)


if __name__ == "__main__":
    main()
