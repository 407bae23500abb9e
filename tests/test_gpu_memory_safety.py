"""Memory safety of the HIP path: bench.py's train steps (the m64 UNet at the headline size, rule-build prefetch and
dWeight side stream on) with every device tensor in its own mapping between unmapped guard pages (tools/guard_alloc.cpp).
An access outside a tensor is a GPU memory access fault and the child process dies; GPU AddressSanitizer is not
available on this pool."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("env", [{"GUARD_MODE": "end"}, {"GUARD_MODE": "start"},
                                 {"GUARD_MODE": "end", "GUARD_POISON": "0xEF"},
                                 {"GUARD_MODE": "end", "GUARD_SYNC": "stream"},
                                 # the dWeight side stream made to lag by ~1 ms per kernel: whatever its kernels read must
                                 # be pinned until they ran (round 3: the rulebooks were not -- the last reference to a
                                 # batch's Metadata goes inside the backward pass, before the streams are joined)
                                 {"GUARD_MODE": "end", "GUARD_SYNC": "stream", "GUARD_DW_DELAY": "2000000"}],
                         ids=["past_the_end", "before_the_start", "poisoned_fresh_memory", "per_stream_frees",
                              "lagging_dweight_stream"])
def test_train_steps_touch_no_byte_outside_their_tensors(env):
    e = dict(os.environ)
    e.update(env)
    p = subprocess.run([sys.executable, "-X", "faulthandler", os.path.join(ROOT, "tools", "guard_run.py"),
                        "--no-cpu-baseline", "--steps", "2", "--warmup", "1"], env=e, cwd=ROOT, timeout=600,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode == 0, (p.returncode, p.stderr[-3000:])
    last = p.stdout.strip().splitlines()[-1]
    assert last.startswith("guard_run:") and last.endswith("clean"), last
