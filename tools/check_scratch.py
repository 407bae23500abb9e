#!/usr/bin/env python3
"""Kernels of libscn_hip.so that use SCRATCH (private memory: register spills, dynamically indexed arrays).
None may: a spill costs HBM traffic in kernels tuned to the register, and a queue's first dispatch with scratch makes
the runtime size that queue's scratch buffer (the rule build runs on its own stream; see scn_rules.hip, sort_keys).
Allowed: the A/B instantiation of rocPRIM's default sort (SCN_SORT_ROCPRIM_DEFAULT=1).
usage: python tools/check_scratch.py [library]      exit code 1 when a kernel uses scratch"""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
ALLOWED = re.compile(r"wrapped_radix_sort_onesweep_configINS0_14default_config")


def kernels(lib):
    """[(kernel name, scratch bytes per lane, vgprs, sgprs)] of every gfx950 code object embedded in `lib`"""
    out = []
    with tempfile.TemporaryDirectory() as tmp:
        copy = os.path.join(tmp, os.path.basename(lib))
        shutil.copy(lib, copy)
        subprocess.check_call([os.path.join(LLVM, "llvm-objdump"), "--offloading", copy], stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.check_output([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], text=True)
            for block in notes.split("- .agpr_count")[1:]:
                def field(key):
                    m = re.search(r"\.%s:\s*(\S+)" % key, block)
                    return m.group(1) if m else None
                out.append((field("name"), int(field("private_segment_fixed_size")), int(field("vgpr_count")),
                            int(field("sgpr_count"))))
    return out


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "occuseg_amd", "libscn_hip.so")
    ks = kernels(lib)
    bad = [(n, s) for n, s, _, _ in ks if s > 0 and not ALLOWED.search(n)]
    for n, s in bad:
        print("SCRATCH %d bytes/lane: %s" % (s, n))
    print("check_scratch: %d kernels, %d with scratch" % (len(ks), len(bad)))
    return 1 if bad or not ks else 0


if __name__ == "__main__":
    sys.exit(main())
