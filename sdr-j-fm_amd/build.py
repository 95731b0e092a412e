"""Build libfmx.so (HIP, gfx950 only) in-tree with hipcc.  No torch, no JIT cache: the .so lands in
sdr-j-fm_amd/lib/ so that it travels with the repository snapshot to the GPU box."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libfmx.so")

# (source, extra flags).  fmx_demod.hip must not contract a*b+c into fma: its LUT index
# expressions reproduce the reference's unfused arithmetic (SURVEY A.14 ii).
SOURCES = [
    ("fmx_front.hip", []),
    ("fmx_front4.hip", []),
    ("fmx_front4lo.hip", []),
    ("fmx_demod.hip", ["-ffp-contract=off"]),
    ("fmx_stageb.hip", ["-ffp-contract=off"]),
    ("fmx_audio.hip", []),
    ("fmx_rds.hip", ["-ffp-contract=off"]),
    ("fmx_ola.hip", ["-ffp-contract=off"]),
    ("fmx_promote.hip", ["-ffp-contract=off"]),
    ("fmx_api.hip", ["-ffp-contract=off"]),
]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libfmx cannot be built (there is no CPU fallback)")
    return exe


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(CSRC, "fmx_front4.hip"))          # (fmx_front4lo.hip is this source compiled a second time)
    headers.append(os.path.join(os.path.dirname(HERE), "include", "fmx.h"))
    objs = []
    cc = hipcc()
    procs = []
    for src, extra in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, src.replace(".hip", ".o"))
        objs.append(o)
        if force or _stale(o, [s] + headers):
            cmd = [cc] + COMMON + extra + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if force or _stale(LIB, objs):
        cmd = [cc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
