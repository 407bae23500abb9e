#!/usr/bin/env python3
"""Measured peaks of the box: f32-MFMA TFLOP/s and HBM stream GB/s (libscn_hip_tuning.so microbenchmarks).
usage: python tools/peaks.py   -> one JSON line (bench.py reports the same numbers as roofline.peak_measured)"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    print(json.dumps(bench.measured_peaks(dev)))
