"""Memory safety of the HIP path: bench.py's train steps (the m64 UNet at the headline size, rule-build prefetch and
dWeight side stream on) with every device tensor in its own mapping between unmapped guard pages (tools/guard_alloc.cpp).
An access outside a tensor is a GPU memory access fault and the child process dies; GPU AddressSanitizer is not
available on this pool."""
import json
import os
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


# (the configurations themselves: tests/guard_chain.py, which tests/conftest.py starts right after collection -- before this
#  process has touched the GPU: a process that has initialised the GPU must not start children on this pool)
@pytest.mark.parametrize("name", ["past_the_end", "before_the_start", "poisoned_fresh_memory", "per_stream_frees",
                                  "lagging_dweight_stream"])
def test_train_steps_touch_no_byte_outside_their_tensors(request, name):
    job = request.config._scn_jobs.get("guard")
    if job is None:
        pytest.skip("no GPU visible at collection time")
    path = os.path.join(job["dir"], name + ".json")
    deadline = time.time() + 3000
    while not os.path.exists(path):
        assert job["proc"].poll() is None or os.path.exists(path), "guard chain ended without a verdict for " + name
        assert time.time() < deadline, "guard chain: no verdict for %s" % name
        time.sleep(2)
    res = json.load(open(path))
    assert res["rc"] == 0, (res["rc"], res["stderr"])
    assert res["last"].startswith("guard_run:") and res["last"].endswith("clean"), res["last"]
