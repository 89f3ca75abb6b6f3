#!/usr/bin/env python3
"""A few launches of the eight-wave and the four-wave GEMM on the bench's shapes, for rocprofv3 --pmc passes (tools/gpu_r5_pmc.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PCLIP_GEMM_4W"] = "0"
from proto_clip_amd import ops  # noqa: E402
from gemm4w_check import case, gemm4w  # noqa: E402

M = 1024 * 197
for name, (m, n, k, act, ub, ur) in {"in_proj": (M, 2304, 768, 0, True, False), "c_fc": (M, 3072, 768, 1, True, False), "c_proj": (M, 768, 3072, 0, True, True),
                                     "sq8192": (8192, 8192, 8192, 0, True, False)}.items():
    a, w, bias, res = case(m, n, k, act, ub, ur)
    out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    for _ in range(3):
        ops.gemm(a, w, bias, act, res, out)
        gemm4w(a, w, bias, act, res, out)
    torch.cuda.synchronize()
