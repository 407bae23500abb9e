"""Build libscn_hip.so (gfx950) in-tree with hipcc.  `python -m occuseg_amd.build`.

Two libraries from the same sources: libscn_hip.so is the product; libscn_hip_tuning.so (-DSCN_TUNING) additionally
holds the kernel-variant knobs, the timing-only ablation instantiations and scn_set_tuning / scn_debug_stamps, and is
loaded only by tools/ and by the variant-equivalence test."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libscn_hip.so")
LIB_TUNING = os.path.join(HERE, "libscn_hip_tuning.so")
SOURCES = ["scn_rules.hip", "scn_dense.hip", "scn_conv.hip", "scn_aux.hip"]
TUNING_ONLY = ["scn_peaks.hip"]  # measured-peak microbenchmarks: tools/peaks.py, bench.py's roofline
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, tuning=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    headers = [os.path.join(CSRC, "scn_common.h"), os.path.join(HERE, "..", "include", "scn_hip.h"),
               os.path.join(CSRC, "tile4_loop_asm.inc"), os.path.join(CSRC, "tile4_loop_asm_tuning.inc")]
    # k_conv_tile4's pass loop in assembly is GENERATED (tools/gen_tile4_loop.py, which also proves on its own instruction list
    # that no instruction touches a register with a load in flight): the committed file must be what the generator writes
    gen = [sys.executable, os.path.join(HERE, "..", "tools", "gen_tile4_loop.py"), "--check"]
    if subprocess.call(gen, stdout=subprocess.DEVNULL if not verbose else None) != 0:
        raise RuntimeError("occuseg_amd/csrc/tile4_loop_asm.inc is stale or fails its checks: python tools/gen_tile4_loop.py")
    sources = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    procs = []
    checkers = []
    objs = {False: [], True: []}
    for tune in ((False, True) if tuning else (False,)):
        for s in sources + (TUNING_ONLY if tune else []):
            src = os.path.join(CSRC, s)
            # only scn_conv.hip has tuning-only code; the other objects are shared
            suffix = ".tuning.o" if tune and s == "scn_conv.hip" else ".o"
            obj = os.path.join(CSRC, s.replace(".hip", suffix))
            objs[tune].append(obj)
            if tune and suffix == ".o" and s not in TUNING_ONLY:
                continue  # compiled in the product pass
            if force or _stale(obj, [src] + headers):
                cmd = [hipcc] + FLAGS + (["-DSCN_TUNING"] if tune else []) + ["-c", src, "-o", obj]
                if verbose:
                    print(" ".join(cmd), flush=True)
                procs.append((cmd, subprocess.Popen(cmd)))
                if s == "scn_conv.hip":
                    # the hand-scheduled kernels issue their loads from inline asm with counted s_waitcnt: prove on the
                    # emitted assembly that no instruction touches a register whose load is still in flight
                    chk = [sys.executable, os.path.join(HERE, "..", "tools", "check_asm_loads.py")] + (["--tuning"] if tune else [])
                    checkers.append((chk, subprocess.Popen(chk, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    for chk, proc in checkers:
        out = proc.communicate()[0].decode()
        if verbose:
            print(out.strip().split("\n")[-1], flush=True)
        if proc.returncode != 0 and not os.environ.get("SCN_SKIP_ASM_CHECK"):  # (debugging builds with printf in a kernel)
            for o in ("scn_conv.o", "scn_conv.tuning.o"):
                if os.path.exists(os.path.join(CSRC, o)):
                    os.remove(os.path.join(CSRC, o))
            raise RuntimeError("check_asm_loads failed:\n" + out)
    for tune, lib in ((False, LIB), (True, LIB_TUNING)):
        if tune and not tuning:
            continue
        if force or procs or _stale(lib, objs[tune]):
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs[tune] + ["-o", lib]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
