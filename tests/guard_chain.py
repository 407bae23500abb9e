"""Started by tests/conftest.py right after collection (before the pytest process touches the GPU): runs tools/guard_run.py once
per configuration of tests/test_gpu_memory_safety.py, one after the other, and leaves <out>/<id>.json = {rc, last, stderr}.
This process itself never initialises the GPU, so starting children from it is safe on this pool."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = [("past_the_end", {"GUARD_MODE": "end"}),
           ("before_the_start", {"GUARD_MODE": "start"}),
           ("poisoned_fresh_memory", {"GUARD_MODE": "end", "GUARD_POISON": "0xEF"}),
           ("per_stream_frees", {"GUARD_MODE": "end", "GUARD_SYNC": "stream"}),
           # the dWeight side stream made to lag by ~1 ms per kernel: whatever its kernels read must be pinned until they ran
           # (round 3: the rulebooks were not -- the last reference to a batch's Metadata goes inside the backward pass, before
           # the streams are joined)
           ("lagging_dweight_stream", {"GUARD_MODE": "end", "GUARD_SYNC": "stream", "GUARD_DW_DELAY": "2000000"})]

if __name__ == "__main__":
    out = sys.argv[1]
    wanted = set(sys.argv[2:])
    for name, env in CONFIGS:
        if wanted and name not in wanted:
            continue
        e = dict(os.environ)
        e.update(env)
        try:
            p = subprocess.run([sys.executable, "-X", "faulthandler", os.path.join(ROOT, "tools", "guard_run.py"),
                                "--no-cpu-baseline", "--steps", "2", "--warmup", "1"], env=e, cwd=ROOT, timeout=600,
                               stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            lines = p.stdout.strip().splitlines()
            res = {"rc": p.returncode, "last": lines[-1] if lines else "", "stderr": p.stderr[-3000:]}
        except subprocess.TimeoutExpired as ex:
            res = {"rc": -1, "last": "timeout", "stderr": str(ex)}
        with open(os.path.join(out, name + ".json.tmp"), "w") as f:
            json.dump(res, f)
        os.replace(os.path.join(out, name + ".json.tmp"), os.path.join(out, name + ".json"))
