import sys, numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import optimum_quanto_amd
from optimum_quanto_amd.library.hip import quanto_hip
from oracle import quanto_oracle as O
from helpers import make_qbits_problem, to_torch, to_numpy
lib = quanto_hip.lib
for dt, N, K in (("fp16", 1024, 1024), ("bf16", 1024, 1024), ("bf16", 1024, 1024), ("bf16", 512, 2048), ("bf16", 2048, 2048)):
    p = make_qbits_problem(K, N, K, dt, group_size=128, zeropoint=False, seed=3)
    x = np.eye(K, dtype=np.float32)
    y = to_numpy(lib.qbits_mm(to_torch(x, dt, "cuda"), torch.from_numpy(p["packed"]).cuda(), to_torch(p["scale"], dt, "cuda"), to_torch(p["shift"], dt, "cuda"),
                              None, 4, 128, N, K, kernel="mfma_large4"))
    w = O.dequantize_qbits_ref(p["packed"], 4, p["scale"], p["shift"], 0, 128, (N, K), dt).astype(np.float32)
    bad = (y != w.T)          # [k, n]
    print(dt, "mismatches", int(bad.sum()), "of", bad.size)
    if bad.any():
        ks, ns = np.nonzero(bad)
        print(" by k-tile (k//64):", np.bincount(ks // 64, minlength=K // 64))
        print(" by k%64 //8:", np.bincount((ks % 64) // 8, minlength=8))
        print(" by k%8:", np.bincount(ks % 8, minlength=8))
        print(" by plane:", np.bincount(ns // (N // 2), minlength=2))
        print(" by n%16:", np.bincount(ns % 16, minlength=16))
        print(" distinct k:", sorted(set(ks.tolist()))[:20], " distinct n//16:", sorted(set((ns // 16).tolist()))[:20])
        for k, n in list(zip(ks, ns))[:3]:
            print("  k", k, "n", n, "got", y[k, n], "want", w[n, k])
