"""Build libscn_hip.so (gfx950) in-tree with hipcc.  `python -m occuseg_amd.build`."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libscn_hip.so")
SOURCES = ["scn_rules.hip", "scn_dense.hip", "scn_conv.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "scn_common.h"), os.path.join(HERE, "..", "include", "scn_hip.h")]
    objs = []
    procs = []
    checker = None
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((cmd, subprocess.Popen(cmd)))
            if s == "scn_conv.hip":
                # k_conv_tile4 schedules its loads by hand (inline asm + counted s_waitcnt): prove on the emitted
                # assembly that no instruction touches a register whose load is still in flight
                chk = [sys.executable, os.path.join(HERE, "..", "tools", "check_asm_loads.py")]
                checker = (chk, subprocess.Popen(chk, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    if checker is not None:
        out = checker[1].communicate()[0].decode()
        if verbose:
            print(out.strip().split("\n")[-1], flush=True)
        if checker[1].returncode != 0:
            os.remove(os.path.join(CSRC, "scn_conv.o"))
            raise RuntimeError("check_asm_loads failed:\n" + out)
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
