"""Build libmmr_hip.so (gfx950) in-tree with hipcc.  No fallbacks: if hipcc is
missing this raises.

Safe under ``torch.distributed.run --nproc-per-node N`` on a clean checkout: the build is serialised by an
fcntl lock in csrc/, objects go to a private temp directory and the library is moved into place with an atomic
``os.replace``, so a rank never links or dlopens a half-written file; ranks that lose the race find the
finished library when they get the lock.  Staleness is decided by a content hash of the sources stored next to
the library (mtimes do not survive the snapshot copy to a GPU box)."""
import fcntl
import hashlib
import os
import shutil
import subprocess
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["api.hip", "tail.hip", "losses.hip", "conv3d.hip", "train.hip", "synth.hip", "eval.hip", "hostio.hip"]
LIB = os.path.join(CSRC, "libmmr_hip.so")
STAMP = LIB + ".srchash"
LOCK = os.path.join(CSRC, ".build.lock")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC"]


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def source_hash():
    """Content hash of everything the library is built from.  include/mmr.h lives at the repo root; a relocated package
    without it still hashes (and loads) its own sources."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    deps = sources() + [os.path.join(CSRC, "common.hpp"), os.path.join(CSRC, "..", "..", "include", "mmr.h")]
    for d in deps:
        if not os.path.exists(d):
            continue
        with open(d, "rb") as f:
            h.update(os.path.basename(d).encode() + b"\0" + f.read())
    return h.hexdigest()


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    with open(STAMP) as f:
        return f.read().strip() != source_hash()


def hipcc_path():
    return os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = hipcc_path()
    with open(LOCK, "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():   # another rank built it while we waited for the lock
                return LIB
            want = source_hash()
            tmp = tempfile.mkdtemp(prefix=".build-", dir=CSRC)
            try:
                objs, procs = [], []
                for s in sources():
                    o = os.path.join(tmp, os.path.basename(s)[:-4] + ".o")
                    objs.append(o)
                    cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
                    if verbose:
                        print(" ".join(cmd))
                    procs.append((cmd, subprocess.Popen(cmd)))
                failed = [cmd for cmd, p in procs if p.wait() != 0]
                if failed:
                    raise RuntimeError("hipcc failed: " + " ".join(failed[0]))
                out = os.path.join(tmp, "libmmr_hip.so")
                subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
                os.replace(out, LIB)
                with open(os.path.join(tmp, "stamp"), "w") as f:
                    f.write(want + "\n")
                os.replace(os.path.join(tmp, "stamp"), STAMP)
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


DIAG_LIB = os.path.join(CSRC, "libmmr_hip_diag.so")


def build_diag(verbose=False):
    """Diagnostic twin of the library (-DMMR_DIAG): adds the cycle-stamped instantiations of the conv / wgrad kernels and
    the mmr_debug_* exports that tools/conv_stamps.py and tools/wgrad_stamps.py read.  Never the measured or shipped
    build; load it with MMR_LIB=<path>."""
    hipcc = hipcc_path()
    tmp = tempfile.mkdtemp(prefix=".build-", dir=CSRC)
    try:
        objs, procs = [], []
        for s in sources():
            o = os.path.join(tmp, os.path.basename(s)[:-4] + ".o")
            objs.append(o)
            cmd = [hipcc] + FLAGS + ["-DMMR_DIAG", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            procs.append((cmd, subprocess.Popen(cmd)))
        failed = [cmd for cmd, p in procs if p.wait() != 0]
        if failed:
            raise RuntimeError("hipcc failed: " + " ".join(failed[0]))
        out = os.path.join(tmp, "libmmr_hip_diag.so")
        subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
        os.replace(out, DIAG_LIB)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return DIAG_LIB


if __name__ == "__main__":
    import sys
    print(build_diag(verbose=True) if "--diag" in sys.argv else build(force=True, verbose=True))
