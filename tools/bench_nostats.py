#!/usr/bin/env python3
"""bench.py with the statistics hand-over switched off (A/B of the forward launches): python tools/bench_nostats.py [bench args]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import occuseg_amd  # noqa: F401
from occuseg_amd.sparseconvnet import conv

conv.EMIT_COLSTATS = False
import bench

bench.main()
