"""Builds liblt_hip.so (gfx950 only) from csrc/*.hip with hipcc.  No torch involved: the library is a
plain C-ABI shared object (include/lt_hip.h)."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "liblt_hip.so")
ARCH = "gfx950"
# -ffp-contract=off: a*b+c is two roundings unless the source says fmaf() -- HIP's __fmul_rn/__fadd_rn are plain operators that
# the compiler would otherwise fuse, which breaks the bit-exact coordinate grid (and makes results depend on inlining)
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-ffp-contract=off"]


def _hipcc():
    h = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(h):
        raise RuntimeError("hipcc not found: cannot build liblt_hip.so")
    return h


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "lt_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    """Compile every csrc/*.hip for gfx950 and link lib/liblt_hip.so.  Returns the library path."""
    if not force and not is_stale():
        return LIB
    if force:
        os.environ["LT_BUILD_FORCE"] = "1"
    try:
        return _build(LIB, "build", [], verbose)
    finally:
        os.environ.pop("LT_BUILD_FORCE", None)


def build_variant(name, defines, verbose=True, only=None):
    """A/B build: lib/liblt_hip_<name>.so compiled with extra -D switches (kernel-scheduling experiments that must be
    compile-time).  Selected at run time with LT_HIP_LIB=<path> (see lt_hip.py); never built or loaded by default.
    only: source basenames the switches concern (e.g. ["conv_igemm3.hip"]); the other objects are taken from the default build
    (which must be up to date) -- a variant then costs one file's compile instead of the whole library's."""
    lib = os.path.join(LIBDIR, "liblt_hip_%s.so" % name)
    return _build(lib, "build_" + name, ["-D" + d for d in defines], verbose, only)


def _build(LIB, objsub, extra, verbose, only=None):
    FLAGS = globals()["FLAGS"] + list(extra)
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(HERE, objsub)
    os.makedirs(objdir, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(HERE, "..", "include", "lt_hip.h")]
    force_all = os.environ.get("LT_BUILD_FORCE") == "1"
    # the flags an object was compiled with travel next to it: a variant rebuilt under the same name with other -D switches recompiles
    stamp = " ".join(FLAGS)

    def compile_one(src):
        if only is not None and os.path.basename(src) not in only:
            base_obj = os.path.join(HERE, "build", os.path.basename(src)[:-4] + ".o")
            if not os.path.exists(base_obj) or os.path.getmtime(base_obj) < os.path.getmtime(src):
                raise RuntimeError("variant build: default object of %s is missing or stale, run lt_build.build() first" % src)
            return base_obj
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        # incremental: an object newer than its source, every header and this script (the flags) is kept
        fl = obj + ".flags"
        same_flags = os.path.exists(fl) and open(fl).read() == stamp
        if not force_all and same_flags and os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(d) for d in [src, __file__] + headers):
            return obj
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr))
        with open(fl, "w") as f:
            f.write(stamp)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        objs = list(ex.map(compile_one, sources()))
    cmd = [hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    if verbose:
        print("built %s (%d sources, %s)" % (LIB, len(objs), ARCH), file=sys.stderr)
    return LIB


if __name__ == "__main__":
    if "--variant" in sys.argv:   # python lt_build.py --variant upfront LT_DMA_UPFRONT
        i = sys.argv.index("--variant")
        build_variant(sys.argv[i + 1], sys.argv[i + 2:])
    else:
        build(force="--force" in sys.argv)
