#!/usr/bin/env python3
"""Structured inputs through qbits_mm_a8 with e4m3 activations: which factor of y = sx * sum_g (s P_g - z A_g) is off?"""
import os, sys
import numpy as np
import torch
os.environ.setdefault("QUANTO_HIP_EXPERIMENT", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from optimum_quanto_amd.library.hip import quanto_hip
from oracle import quanto_oracle as O
from helpers import fp8_tensor, to_torch, to_numpy

DEV = "cuda"
def run(q, a_codes, s, z, dt="fp16", bm="64"):
    os.environ["QUANTO_HIP_A8_BM"] = bm
    N, K = q.shape
    packed = O.pack_weights(O.group(q.astype(np.uint8), 0, 128), 4)
    G = K // 128
    scale = O.round_to(np.full((N * G, 1), s, np.float32), dt)
    shift = O.round_to(np.full((N * G, 1), z, np.float32), dt)
    y = quanto_hip.lib.qbits_mm_a8(fp8_tensor(a_codes, "e4m3fn", DEV), to_torch(np.array([1.0], np.float32), dt, DEV), torch.from_numpy(packed).to(DEV),
                                   to_torch(scale, dt, DEV), to_torch(shift, dt, DEV), None, 4, 128, N, K)
    return to_numpy(y).astype(np.float64)

M, N, K = 64, 128, 256
one = np.full((M, K), 0x38, np.uint8)
for c in (0, 1, 2, 3, 7, 8, 15):
    y = run(np.full((N, K), c), one, 1.0, 0.0)
    print(f"q={c:2d} a=1 s=1 z=0: expect {K*c}, got min {y.min()} max {y.max()}")
y = run(np.zeros((N, K)), one, 1.0, 1.0)
print(f"q=0 a=1 s=1 z=1: expect {-K}, got min {y.min()} max {y.max()}")
y = run(np.ones((N, K)), one, 1.0, 1.0)
print(f"q=1 a=1 s=1 z=1: expect 0, got min {y.min()} max {y.max()}")
rng = np.random.default_rng(0)
codes = O.fp8_encode(rng.integers(-8, 9, size=(M, K)).astype(np.float32), "e4m3fn")
av = O.fp8_decode(codes, "e4m3fn").astype(np.float64)
q = rng.integers(0, 16, size=(N, K))
for bm in ("64", "128"):
    y = run(q, codes, 1.0, 0.0, bm=bm)
    want = av @ q.T.astype(np.float64)
    bad = np.argwhere(y != want)
    print(f"bm={bm} random ints z=0: mismatches {len(bad)} of {y.size}; first {bad[:5].tolist()}; y[0,:4]={y[0,:4]} want {want[0,:4]}")
    y = run(np.zeros((N, K)), codes, 1.0, 1.0, bm=bm)
    want = -av.sum(1, keepdims=True) * np.ones((1, N))
    bad = np.argwhere(y != want)
    print(f"bm={bm} q=0 z=1 (row sums): mismatches {len(bad)}; y[:4,0]={y[:4,0]} want {want[:4,0]}")
    # one group only
    y = run(q[:, :128], codes[:, :128], 1.0, 0.0, bm=bm)
    want = av[:, :128] @ q[:, :128].T.astype(np.float64)
    print(f"bm={bm} K=128 z=0: mismatches {(y != want).sum()}")
    y = run(np.zeros((N, 128)), codes[:, :128], 1.0, 1.0, bm=bm)
    want = -av[:, :128].sum(1, keepdims=True) * np.ones((1, N))
    print(f"bm={bm} K=128 q=0 z=1 (row sums): mismatches {(y != want).sum()}; y[:4,0]={y[:4,0]} want {want[:4,0]}")
