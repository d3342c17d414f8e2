"""Build libmmr_hip.so (gfx950) in-tree with hipcc.  No fallbacks: if hipcc is
missing this raises."""
import os
import subprocess

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
SOURCES = ["api.hip", "tail.hip", "losses.hip", "conv3d.hip", "train.hip", "synth.hip", "eval.hip"]
LIB = os.path.join(CSRC, "libmmr_hip.so")


def sources():
    return [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, "common.hpp"),
                        os.path.join(CSRC, "..", "..", "include", "mmr.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in sources():
        o = s[:-4] + ".o"
        objs.append(o)
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-fPIC", "-c", s, "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
