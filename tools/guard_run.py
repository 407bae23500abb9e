#!/usr/bin/env python3
"""Memory-safety check of the HIP path: bench.py's train steps with every device tensor in its own mapping between
unmapped guard ranges (tools/guard_alloc.cpp; GPU AddressSanitizer is not available on this pool).  An access outside
any tensor -- past its end with GUARD_MODE=end, before its start with GUARD_MODE=start -- is a GPU memory access fault;
with HIP_LAUNCH_BLOCKING=1 (set here) the host is then inside the faulting launch and `python -X faulthandler` prints
where.  Exit code 0 and the line "guard_run: ... clean" mean no launch of the steps touched a byte outside its operands.

GUARD_SYNC=stream: launches stay asynchronous and a free waits only for the tensor's allocation stream and the streams named
by Tensor.record_stream -- what a stream-ordered caching allocator knows; a kernel on another stream that still uses the
tensor then faults (a missing record_stream / event between two streams).  GUARD_LOG=1 lists every allocation.

usage: GUARD_MODE=end|start python -X faulthandler tools/guard_run.py [bench.py arguments; default: 2 warm-up + 2 steps]
       (run it under `timeout`: a fault may leave the process hanging)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BY_STREAM = os.environ.get("GUARD_SYNC") == "stream"
if not BY_STREAM:
    os.environ.setdefault("HIP_LAUNCH_BLOCKING", "1")
    os.environ.setdefault("AMD_SERIALIZE_KERNEL", "3")


def build():
    so = os.path.join(ROOT, "tools", "libguard_alloc.so")
    src = os.path.join(ROOT, "tools", "guard_alloc.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["hipcc", "-shared", "-fPIC", "-O1", "-o", so, src])
    return so


def main():
    so = build()
    if "--build-only" in sys.argv:
        return
    import torch
    alloc = torch.cuda.memory.CUDAPluggableAllocator(so, "guard_malloc", "guard_free")
    torch.cuda.memory.change_current_allocator(alloc)
    if BY_STREAM:  # (the pluggable allocator is not told about record_stream: tell it)
        glib = ctypes.CDLL(so)
        glib.guard_record.argtypes = [ctypes.c_void_p, ctypes.c_void_p]

        def record_stream(self, stream):
            if self.numel() and glib.guard_record(self.untyped_storage().data_ptr(), stream.cuda_stream):
                raise RuntimeError("record_stream: tensor not from the guard allocator")
        torch.Tensor.record_stream = record_stream
    if os.environ.get("GUARD_DW_DELAY"):  # (tests: the dWeight side stream lags the main stream, see SCN.DW_SIDE_TEST_DELAY_CYCLES)
        from occuseg_amd import SCN
        SCN.DW_SIDE_TEST_DELAY_CYCLES = int(os.environ["GUARD_DW_DELAY"])
    import bench
    sys.argv = [os.path.join(ROOT, "bench.py")] + (sys.argv[1:] or ["--no-cpu-baseline", "--steps", "2", "--warmup", "2"])
    bench.main()
    torch.cuda.synchronize()
    lib = ctypes.CDLL(so)
    c, b, live = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    lib.guard_stats(ctypes.byref(c), ctypes.byref(b), ctypes.byref(live))
    print("guard_run: mode %s, %d allocations (%.1f GiB mapped in total, %d live at exit): clean"
          % (os.environ.get("GUARD_MODE", "end"), c.value, b.value / 2.0 ** 30, live.value), flush=True)


if __name__ == "__main__":
    main()
