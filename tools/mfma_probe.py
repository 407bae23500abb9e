#!/usr/bin/env python3
"""What a wave's instruction stream around its f32 MFMAs costs (libscn_hip_tuning.so: k_mfma_probe).
usage: python tools/mfma_probe.py"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from occuseg_amd import _lib

MODES = ["2 accumulators, back to back", "8 accumulators", "2 acc, operands from 32 + 16 registers", "+ 1 v_add per MFMA", "+ 3 v_add per MFMA",
         "+ 1 s_add per MFMA", "+ scalar branch (not taken) per 8 MFMAs", "+ ds_read_b128 per 4 MFMAs", "+ EXEC=0 buffer_load per 4 MFMAs",
         "+ L2-hit buffer_load_dwordx4 per 4 MFMAs", "+ 1 v_accvgpr_read per MFMA", "+ 1 v_mov_b64 per MFMA", "+ 1 v_mul_lo_u32 per MFMA",
         "+ 1 v_mad_u32_u24 per MFMA", "+ 1 ds_add_f32 per MFMA", "+ 1 v_readfirstlane per MFMA", "+ 1 v_pk_add_f32 per MFMA", "+ 1 v_cndmask per MFMA",
         "+ 1 s_nop 0 per MFMA", "+ ds_write_b128 of an accumulator per 4 MFMAs",
         "64 VALU per 64 MFMAs: 2 after every 2nd", "64 per 64: 4 after every 4th", "64 per 64: 16 after every 16th", "64 per 64: 64 after the 64th",
         "32 per 64: 1 after every 2nd", "16 per 64: 1 after every 4th", "16 per 64: 16 after the 64th", "16 v_pk_add_f32 per 64, after the 64th",
         "asm MFMA, accumulators in VGPRs", "asm MFMA, accumulators in AGPRs",
         "acc in VGPRs + buffer_load_dwordx4 per 4 MFMAs", "acc in AGPRs + buffer_load_dwordx4 per 4 MFMAs",
         "acc in VGPRs + buffer_load_dwordx4 per 2 MFMAs", "acc in AGPRs + buffer_load_dwordx4 per 2 MFMAs",
         "asm v_add_f32_e32 f, 1.0, f (row 3 as asm)", "asm v_add_f32_e32 f, a[], f (reads an MFMA operand)",
         "asm v_add_f32_e32 f, g, f (free registers)", "asm v_add_f32_e64 f, g, f (VOP3)", "asm v_add_u32_e32 f, g, f",
         "asm v_lshl_add_u32 f, g, 2, f (VOP3)", "asm v_pk_add_f32, free registers", "asm v_mad_u32_u24, free registers",
         "compiler f += a[] (reads an MFMA operand)", "compiler pair f += g", "asm v_mov_b32 f, g (free registers)",
         "asm v_pk_add_f32 reading an idle accumulator"]
ONLY = [int(x) for x in sys.argv[1].split(",")] if len(sys.argv) > 1 else None

if __name__ == "__main__":
    dev = torch.device("cuda:0")
    lib = _lib.use_tuning_build()
    scratch = torch.zeros(1024, device=dev)
    src = torch.randn(1 << 18, device=dev)
    out = ctypes.c_double(0.0)
    for mode, name in enumerate(MODES):
        if ONLY is not None and mode not in ONLY:
            continue
        line = "%-45s" % name
        for wps in (1, 2, 3, 4):
            _lib.check(lib.scn_mfma_probe(scratch.data_ptr(), src.data_ptr(), mode, wps, ctypes.byref(out), _lib.stream()))
            line += "  %dw/SIMD %6.1f TF" % (wps, out.value)
        print(line, flush=True)
