#!/usr/bin/env python3
"""int4 prefill: which kernel wins at which M.  Times quanto_hip.lib.qbits_mm with an explicit kernel (fused int4 GEMM with both
token-tile heights, dequantize + dense GEMM, streaming kernel) over a grid of shapes; hipGraph of 20 calls, interleaved rounds.

    python scripts/ab_prefill.py [--shapes 4096x4096 14336x4096 4096x14336] [--ms 256 512 1024 2048 4096]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

os.environ.setdefault("QUANTO_HIP_EXPERIMENT", "1")  # the library reads its knobs only behind this switch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="+", default=["4096x4096", "14336x4096", "4096x14336"], help="NxK")
    ap.add_argument("--ms", type=int, nargs="+", default=[256, 512, 1024, 2048, 4096])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--variants", nargs="*", default=None)
    ap.add_argument("--fused-env", nargs="*", default=None, help='knob sets for mfma_fused4, e.g. "BM=64,SPLIT=1" "BM=64,SPLIT=1,ABLATE=1" ("" = defaults)')
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    import optimum_quanto_amd  # noqa: F401
    from optimum_quanto_amd.library.hip import quanto_hip

    lib = quanto_hip.lib
    B, S = "QUANTO_HIP_FUSED4_BM", "QUANTO_HIP_FUSED4_SPLIT"
    variants = [("mfma_fused4", {}), ("mfma_fused4", {B: "64", S: "1"}), ("mfma_fused4", {B: "64", S: "2"}), ("mfma_fused4", {B: "64", S: "4"}),
                ("mfma_fused4", {B: "64", S: "8"}), ("mfma_fused4", {B: "128", S: "1"}), ("dequant_mfma", {}), ("skinny", {})]
    if args.variants:
        variants = [v for v in variants if v[0] in args.variants]
    if args.fused_env is not None:  # explicit list of knob settings for the fused kernel: "BM=64,SPLIT=1,ABLATE=1" ...
        variants = [v for v in variants if v[0] != "mfma_fused4"]
        for spec in args.fused_env:
            variants.append(("mfma_fused4", {"QUANTO_HIP_FUSED4_" + kv.split("=")[0]: kv.split("=")[1] for kv in spec.split(",") if kv}))
    for shape in args.shapes:
        N, K = (int(v) for v in shape.split("x"))
        g = torch.Generator(device=dev).manual_seed(0)
        w = (torch.randn((N, K), generator=g, device=dev) * 0.02).to(torch.bfloat16).float()
        packed, scale, shift = bench.quantize_int4(w)
        del w
        for M in args.ms:
            x = torch.randn((M, K), generator=g, device=dev).to(torch.bfloat16)
            graphs = {}
            for kernel, env in variants:
                if kernel == "skinny" and M > 256:
                    continue
                os.environ.update(env)
                try:
                    call = lambda: lib.qbits_mm(x, packed, scale, shift, None, 4, 128, N, K, kernel=kernel)  # noqa: E731
                    call()
                    torch.cuda.synchronize()
                    side = torch.cuda.Stream()
                    side.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(side):
                        gr = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(gr, stream=side):
                            for _ in range(20):
                                call()
                    torch.cuda.current_stream().wait_stream(side)
                    gr.replay()
                    torch.cuda.synchronize()
                    graphs[kernel + "".join(f"[{k.split('_')[-1]}={v}]" for k, v in env.items())] = gr
                except Exception as e:  # a kernel that does not support the shape
                    print(json.dumps({"N": N, "K": K, "M": M, "kernel": kernel, "error": repr(e)[:120]}), flush=True)
                for k in env:
                    os.environ.pop(k, None)
            for gr in graphs.values():
                for _ in range(3):
                    gr.replay()
            torch.cuda.synchronize()
            times = {k: [] for k in graphs}
            for _ in range(args.rounds):
                for k, gr in graphs.items():
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    gr.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    times[k].append(e0.elapsed_time(e1) * 1e3 / 20)
            print(json.dumps({"N": N, "K": K, "M": M, **{k: round(float(np.median(v)), 2) for k, v in times.items()}}), flush=True)
            del graphs


if __name__ == "__main__":
    main()
