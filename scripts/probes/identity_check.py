#!/usr/bin/env python3
"""Identity-activation check of the int4 kernels that dequantize between MFMAs (profiles/r04_packed_fp32_next_to_mfma.md): with x = I the
product IS the operand matrix the kernel built, compared bit for bit with the oracle's dequantized weight, over many launches.

    python scripts/probes/identity_check.py [--lib path/to/libquanto_hip.so] [--launches 60]

--lib loads another build of the library (e.g. scripts/probes/libquanto_hip_slp.so: the same sources WITHOUT -fno-slp-vectorize on
qbits_mfma_large.hip / qconv_mfma.hip / qmm_mfma.hip, i.e. with hipcc's packed fp32 forms back in)."""
import argparse
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import optimum_quanto_amd  # noqa: E402,F401
from helpers import make_qbits_problem, to_numpy, to_torch  # noqa: E402
from optimum_quanto_amd.library import hip as H  # noqa: E402
from oracle import quanto_oracle as O  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lib", default=None)
    ap.add_argument("--launches", type=int, default=60)
    args = ap.parse_args()
    lib = H._Bindings(ctypes.CDLL(os.path.abspath(args.lib))) if args.lib else H.quanto_hip.lib
    tag = os.path.basename(args.lib) if args.lib else "product build"
    for dt, N, K in (("bf16", 1024, 1024), ("fp16", 1024, 1024), ("bf16", 2048, 2048)):
        p = make_qbits_problem(K, N, K, dt, group_size=128, zeropoint=False, seed=3)
        x = to_torch(np.eye(K, dtype=np.float32), dt, "cuda")
        packed, scale, shift = torch.from_numpy(p["packed"]).cuda(), to_torch(p["scale"], dt, "cuda"), to_torch(p["shift"], dt, "cuda")
        w = O.dequantize_qbits_ref(p["packed"], 4, p["scale"], p["shift"], 0, 128, (N, K), dt).astype(np.float32)
        bad_launches, bad_elems, lanes = 0, 0, np.zeros(4, dtype=np.int64)
        for _ in range(args.launches):
            y = to_numpy(lib.qbits_mm(x, packed, scale, shift, None, 4, 128, N, K, kernel="mfma_large4"))
            bad = y != w.T
            if bad.any():
                bad_launches += 1
                bad_elems += int(bad.sum())
        print(json.dumps({"lib": tag, "kernel": "mfma_large4", "dtype": dt, "N": N, "K": K, "launches": args.launches, "launches_with_wrong_elements": bad_launches,
                          "wrong_elements": bad_elems}), flush=True)


if __name__ == "__main__":
    main()
