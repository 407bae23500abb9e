#!/usr/bin/env python3
"""Compare the output of forward-kernel variants on the bench scenes (tuning tool).
usage: python tools/check_variant.py [--variants 2,4]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occuseg_amd import SCN, _lib, scenes


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="2,4")
    ap.add_argument("--sites", type=int, default=40000)
    ap.add_argument("--dbg", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = _lib.use_tuning_build()
    coords, _ = scenes.batch([1000, 1001], args.sites)
    md = SCN.Metadata_3()
    size = torch.tensor([4096] * 3)
    md.inputLayer(size, torch.from_numpy(coords), 0, 4, dev)
    f3 = torch.tensor([3, 3, 3])
    rb = md.getSubmanifoldRuleBook(size, f3)
    n = md.getNActive(size)
    vs = [int(v) for v in args.variants.split(",")]
    for (ci, co) in ((64, 64), (128, 64), (192, 128), (64, 192)):
        x = torch.randn(n, ci, device=dev)
        w = torch.randn(27, ci, co, device=dev) * 0.05
        b = torch.randn(co, device=dev)
        outs = []
        for v in vs:
            lib.scn_set_tuning(0, v)
            SCN.invalidate_weight_cache()  # variant 1 packs the weights in a different layout
            lib.scn_set_tuning(1, args.dbg if v == 4 else 0)
            out = torch.empty(0, device=dev)
            SCN._conv(x, out, w, b, rb, 0, 1, n)
            outs.append(out.clone())
        ref = outs[0]
        for v, o in zip(vs[1:], outs[1:]):
            err = (o - ref).abs().max().item() / ref.abs().max().item()
            print("N=%d %d->%d v%d vs v%d: rel err %.3g %s" % (n, ci, co, v, vs[0], err, "OK" if err < 1e-5 else "MISMATCH"), flush=True)
    lib.scn_set_tuning(0, 4)
    lib.scn_set_tuning(1, 0)


if __name__ == "__main__":
    main()
