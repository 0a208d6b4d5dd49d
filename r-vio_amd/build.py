"""In-tree build of librvio_hip.so (hand-written HIP for gfx950, one hipcc call).

hipcc cross-compiles without a GPU, so this runs in the CPU-only container; the
built .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "librvio_hip.so")


def sources():
    """Everything the one hipcc call reads: every file under csrc/ (rvio_hip.hip includes the other kernel files) + the C header."""
    fs = [os.path.join(CSRC, f) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".h", ".inc"))]
    return fs + [os.path.join(HERE, "..", "include", "rvio_hip.h")]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(f) > t for f in sources())


def build(force=False, verbose=False):
    """Compile csrc/ for gfx950 if the library is missing or older than a source."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value"] + \
          os.environ.get("RVIO_HIPCC_FLAGS", "").split() + [os.path.join(CSRC, "rvio_hip.hip"), "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
