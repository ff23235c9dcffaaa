#!/usr/bin/env python3
"""Runs INSIDE a subprocess of the CPU test-suite, in the build container only: imports the REAL reference
(/root/reference, copied to a scratch directory because importing it writes build artefacts into its tree) and checks

  cpu_path : oracle/reference_cpu_path.py issues the same arithmetic as the reference's CPU QLinear path (torch.equal);
  plugin   : importing optimum_quanto_amd AFTER optimum.quanto installs the backend into the reference.

Prints one line per check and ``ALL-OK`` at the end; any assertion error is the test failure."""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.environ.get("QUANTO_REFERENCE") or ("/root/reference" if os.path.isdir("/root/reference") else os.path.join(ROOT, ".refcopy"))


def import_reference():
    scratch = tempfile.mkdtemp(prefix="quanto_ref_")
    dst = os.path.join(scratch, "ref")
    shutil.copytree(REF, dst, ignore=shutil.ignore_patterns(".git", "*.png", "bench", "examples"))
    sys.path.insert(0, dst)
    import optimum.quanto  # noqa: F401

    return scratch


def build_weights(torch, Q):
    torch.manual_seed(0)
    out = {}
    for dt in (torch.float32, torch.bfloat16):
        w = (torch.randn(96, 256) * 0.02).to(dt)
        x = torch.randn(5, 256).to(dt)
        bias = torch.randn(96).to(dt)
        q8 = Q.quantize_weight(w, Q.qint8, 0, Q.AbsmaxOptimizer()(w, Q.qint8, 0), optimized=False)
        f8 = Q.quantize_weight(w, Q.qfloat8_e4m3fn, 0, Q.AbsmaxOptimizer()(w, Q.qfloat8_e4m3fn, 0), optimized=False)
        s4, z4 = Q.MaxOptimizer()(w, Q.qint4, 0, 128)
        q4 = Q.quantize_weight(w, Q.qint4, 0, s4, z4, group_size=128, optimized=False)
        q4o = Q.quantize_weight(w, Q.qint4, 0, s4, z4, group_size=128, optimized=True)  # bf16 on CPU -> TinyGemm subclass
        out[dt] = dict(w=w, x=x, bias=bias, q8=q8, f8=f8, q4=q4, q4o=q4o)
    return out


def check_cpu_path():
    scratch = import_reference()
    import torch
    import optimum.quanto as Q

    sys.path.insert(0, ROOT)
    from oracle import reference_cpu_path as R

    F = torch.nn.functional.linear
    for dt, t in build_weights(torch, Q).items():
        x, bias = t["x"], t["bias"]
        with torch.no_grad():
            # 8-bit: int8 (bf16 takes _weight_int8pack_mm, fp32 the generic path) and fp8
            for key in ("q8", "f8"):
                qw = t[key]
                assert torch.equal(F(x, qw, bias), R.qbytes_linear(x, qw._data, qw._scale, bias)), (key, dt)
                assert torch.equal(torch.ops.quanto.qbytes_mm(x, qw._data, qw._scale), R.qbytes_mm_cpu(x, qw._data, qw._scale)), (key, dt)
            # int8 x int8
            a8 = torch.randint(-127, 127, (24, 256), dtype=torch.int8)
            sc = (t["q8"]._scale * 0.01)
            assert torch.equal(torch.ops.quanto.qbytes_mm(a8, t["q8"]._data, sc), R.qbytes_mm_cpu(a8, t["q8"]._data, sc)), ("i8i8", dt)
            # int4 generic class
            qw = t["q4"]
            assert type(qw).__name__ == "WeightQBitsTensor"
            assert torch.equal(qw.dequantize(), R.dequantize_qbits(qw._data._data, qw._scale, qw._shift, 4, 128, 96, 256)), ("dq4", dt)
            assert torch.equal(F(x, qw, bias), R.qbits_linear_generic(x, qw._data._data, qw._scale, qw._shift, 4, 128, 96, 256, bias)), ("q4", dt)
            # int4 optimized class on CPU: TinyGemm when the scales are bf16
            qo = t["q4o"]
            if dt == torch.bfloat16:
                assert type(qo).__name__ == "TinyGemmWeightQBitsTensor", type(qo).__name__
                data, scale_shift = R.tinygemm_pack(qw._data._data, qw._scale, qw._shift, 128, 96, 256)
                assert torch.equal(data, qo._data._data) and torch.equal(scale_shift, qo._scale_shift)
                assert torch.equal(F(x, qo, bias), R.tinygemm_linear(x, data, 128, scale_shift, 96, bias)), ("tinygemm", dt)
        print(f"cpu_path {dt}: ok")
    shutil.rmtree(scratch, ignore_errors=True)


def check_plugin():
    scratch = import_reference()
    import torch
    import optimum.quanto as Q
    from optimum.quanto.library.extensions import get_extension, is_extension_available

    F = torch.nn.functional.linear
    ws = build_weights(torch, Q)
    with torch.no_grad():
        before = {dt: {k: F(t["x"], t[k], t["bias"]) for k in ("q8", "f8", "q4")} for dt, t in ws.items()}
    had_qbits_mm = hasattr(torch.ops.quanto, "qbits_mm")

    sys.path.insert(0, ROOT)
    import optimum_quanto_amd  # noqa: F401  -> plug-in mode
    from optimum_quanto_amd.library import plugin
    from optimum_quanto_amd.library.hip import quanto_hip

    assert plugin.installed(), "plug-in mode did not engage"
    assert not had_qbits_mm
    for op in ("qbits_mm", "dequantize_qbits", "qbits_mm_multi", "qbytes_mm_bias"):
        assert hasattr(torch.ops.quanto, op), f"quanto::{op} was not added"
    print("plugin: new ops defined")
    # the ROCm ("CUDA") kernels of the reference's own ops now come from this backend
    ops_py = os.path.join("optimum_quanto_amd", "library", "ops.py")
    for op in ("unpack", "qbytes_mm", "quantize_symmetric", "quantize_affine"):
        dump = torch._C._dispatch_dump(f"quanto::{op}")
        cuda = [ln for ln in dump.splitlines() if ln.split(":")[0].strip() == "CUDA"]  # the ACTIVE kernel; "CUDA (inactive)" = the overridden one
        assert cuda and all(ops_py in ln for ln in cuda), f"quanto::{op} CUDA kernel is not ours:\n{dump}"
        # and the reference's CPU / default registrations are untouched
        others = [ln for ln in dump.splitlines() if ln.split(":")[0].strip() in ("CPU", "CompositeExplicitAutograd[alias]", "CompositeExplicitAutograd")]
        assert others and not any(ops_py in ln for ln in others), dump
    print("plugin: CUDA kernels overridden, CPU registrations untouched")
    # extension registry of the reference
    assert is_extension_available("quanto_hip") and get_extension("quanto_hip") is quanto_hip
    assert get_extension("quanto_hip").name == "quanto_hip" and get_extension("quanto_hip").cdll is not None
    assert is_extension_available("quanto_cpp")
    print("plugin: get_extension('quanto_hip') resolves to libquanto_hip.so")
    # CPU behaviour of the reference unchanged (the patched __torch_function__ only routes ROCm tensors)
    with torch.no_grad():
        for dt, t in ws.items():
            for k in ("q8", "f8", "q4"):
                assert torch.equal(before[dt][k], F(t["x"], t[k], t["bias"])), (dt, k)
            # what a ROCm tensor would be routed to, run here through the op's default (CPU) implementation
            y = plugin.fused_qbits_linear(t["x"], t["q4"], t["bias"])
            assert torch.equal(y, before[dt]["q4"]), ("fused route", dt)
    print("plugin: reference CPU results unchanged; fused route equals the reference")
    # gradients still flow through the routed function (straight-through backward of the reference)
    t = ws[torch.float32]
    x = t["x"].clone().requires_grad_(True)
    plugin.fused_qbits_linear(x, t["q4"], t["bias"]).sum().backward()
    xr = t["x"].clone().requires_grad_(True)
    F(xr, t["q4"], t["bias"]).sum().backward()
    assert torch.equal(x.grad, xr.grad)
    # the reference's own QLinear / quantize / freeze drive it
    lin = torch.nn.Linear(256, 96)
    Q.quantize(lin, weights=Q.qint4)
    print("plugin: backward ok")
    shutil.rmtree(scratch, ignore_errors=True)


def check_plugin_gpu():
    """On the MI355X box, with a shipped scratch copy of the reference: a ROCm tensor goes through the reference's own classes
    (QLinear / quantize / freeze / WeightQBitsTensor.__torch_function__) and lands on this library's fast kernels."""
    scratch = import_reference()
    import torch
    import optimum.quanto as Q

    assert torch.cuda.is_available() and torch.version.hip is not None
    sys.path.insert(0, ROOT)
    import optimum_quanto_amd  # noqa: F401  -> plug-in mode
    from optimum_quanto_amd.library import plugin
    from optimum_quanto_amd.library.hip import quanto_hip
    from optimum.quanto.library.extensions import get_extension

    assert plugin.installed() and get_extension("quanto_hip") is quanto_hip
    lib = quanto_hip.lib
    F = torch.nn.functional.linear
    torch.manual_seed(0)
    fast = {1: ("gemv",), 8: ("mmv", "skinny"), 48: ("skinny",), 100: ("mfma_fused4",), 512: ("mfma_fused4",), 2048: ("dequant_mfma", "mfma_fused4")}
    for dt in (torch.bfloat16, torch.float16):
        N, K = 1024, 1024
        w = (torch.randn(N, K) * 0.02).to(dt)
        s4, z4 = Q.MaxOptimizer()(w, Q.qint4, 0, 128)
        q4 = Q.quantize_weight(w, Q.qint4, 0, s4, z4, group_size=128, optimized=False)
        assert type(q4).__name__ == "WeightQBitsTensor"
        q4d = q4.to("cuda")
        assert type(q4d).__name__ == "WeightQBitsTensor" and q4d.device.type == "cuda"
        assert torch.equal(q4d.dequantize().cpu(), q4.dequantize())  # cross-device equality of the generic class (the reference's own gate)
        for M, kernels in fast.items():
            x = torch.randn(M, K).to(dt)
            with torch.no_grad():
                want = F(x.float(), q4.dequantize().float())   # fp32 CPU reference on the same integers / scales
                got = F(x.cuda(), q4d)
            assert lib.last_kernel() in kernels, (M, lib.last_kernel())
            err = (got.float().cpu() - want).abs().max().item() / want.abs().max().item()
            assert err < 2e-2, (dt, M, err)                    # the reference's own tolerance on cuda (weight_helpers.py:19-37)
        # 8-bit: the reference calls quanto::qbytes_mm itself; the CUDA kernel behind it is ours now
        q8 = Q.quantize_weight(w, Q.qint8, 0, Q.AbsmaxOptimizer()(w, Q.qint8, 0), optimized=False).to("cuda")
        with torch.no_grad():
            got = F(torch.randn(1, K).to(dt).cuda(), q8)
        assert lib.last_kernel() == "gemv", lib.last_kernel()
        print(f"plugin_gpu {dt}: int4 M in {sorted(fast)} on {sorted(set(sum(fast.values(), ())))}; int8 decode on gemv")
    # the reference's module API end to end on the device
    model = torch.nn.Sequential(torch.nn.Linear(1024, 512, bias=True)).to(torch.bfloat16).cuda()
    x = torch.randn(3, 1024, device="cuda", dtype=torch.bfloat16)
    ref = model(x)
    Q.quantize(model, weights=Q.qint4)
    Q.freeze(model)
    qmods = [m for m in model.modules() if isinstance(m, Q.nn.QLinear)]
    assert len(qmods) == 1 and type(qmods[0].weight).__name__ == "WeightQBitsTensor"
    with torch.no_grad():
        y = model(x)
    assert lib.last_kernel() == "gemv", lib.last_kernel()
    assert (y.float() - ref.float()).abs().max().item() / ref.float().abs().max().item() < 0.2  # int4 quantization error only
    print("plugin_gpu: reference QLinear(qint4).forward ->", lib.last_kernel(), type(qmods[0]).__name__)
    shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    {"cpu_path": check_cpu_path, "plugin": check_plugin, "plugin_gpu": check_plugin_gpu}[sys.argv[1]]()
    print("ALL-OK")
