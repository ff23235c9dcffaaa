"""BASELINE config 5 at test size: a random-init Llama with qint4 Linears (lm_head excluded).

CPU part: quantize -> generate -> save_pretrained -> from_pretrained round trip (what the reference checks in
tests/models/test_quantized_model_for_causal_lm.py with hub models).  GPU part: the same model moved to the device must
produce the CPU path's logits within the reference's own tolerance, with every quantized Linear running a HIP kernel.
"""
import pytest
import torch

import optimum_quanto_amd as Q


def tiny_llama(dtype=torch.float32, seed=0):
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(seed)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=512, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, vocab_size=320, max_position_embeddings=128, tie_word_embeddings=False)
    return LlamaForCausalLM(cfg).to(dtype).eval()


def test_quantize_generate_save_reload(tmp_path):
    model = tiny_llama()
    qmodel = Q.QuantizedModelForCausalLM.quantize(model, weights=Q.qint4, exclude="lm_head")
    linears = [m for m in model.modules() if isinstance(m, Q.QLinear)]
    assert len(linears) == 2 * 7 and all(m.frozen for m in linears)
    assert not isinstance(model.lm_head, Q.QLinear)
    ids = torch.randint(1, 319, (2, 8))
    with torch.no_grad():
        logits = qmodel(input_ids=ids).logits
        out = qmodel.generate(input_ids=ids, max_new_tokens=4, do_sample=False)
    assert out.shape == (2, 12)
    qmodel.save_pretrained(tmp_path)
    import json
    cfg = json.load(open(tmp_path / "config.json"))  # the compute dtype travels in config.json under the name this transformers serialises
    assert "float32" in (str(cfg.get("dtype")), str(cfg.get("torch_dtype")))
    again = Q.QuantizedModelForCausalLM.from_pretrained(tmp_path)
    assert isinstance(again.model.layers[0].self_attn.q_proj.weight, Q.WeightQBitsTensor)
    with torch.no_grad():
        assert torch.equal(again(input_ids=ids).logits, logits)


@pytest.mark.gpu
@pytest.mark.parametrize("weights", ["qint4", "qint8"])
def test_tiny_llama_on_device_matches_cpu_path(weights):
    model = tiny_llama(torch.bfloat16)
    qmodel = Q.QuantizedModelForCausalLM.quantize(model, weights=weights, exclude="lm_head")
    ids = torch.randint(1, 319, (3, 16))
    with torch.no_grad():
        ref = qmodel(input_ids=ids).logits.float()
    model.to("cuda")
    from optimum_quanto_amd.library.hip import quanto_hip
    with torch.no_grad():
        got = qmodel(input_ids=ids.to("cuda")).logits.float().cpu()
        assert quanto_hip.lib.last_kernel() in ("mfma", "gemv", "mfma_large", "skinny")
        step = qmodel(input_ids=ids[:1, :1].to("cuda")).logits  # decode-shaped call -> GEMV kernels
        assert quanto_hip.lib.last_kernel() == "gemv" and torch.isfinite(step.float()).all()
    cos = torch.nn.functional.cosine_similarity(got.flatten(), ref.flatten(), dim=0)
    assert cos > 0.999, float(cos)
    assert (got - ref).abs().max() / ref.abs().max() < 5e-2


def test_sharded_checkpoint_round_trip(tmp_path):
    """Shards + index on disk (the reference's sharded format, models/shared_dict.py:22-53) reload to identical logits."""
    import os

    model = tiny_llama()
    qmodel = Q.QuantizedModelForCausalLM.quantize(model, weights=Q.qint4, exclude="lm_head")
    ids = torch.randint(1, 319, (2, 8))
    with torch.no_grad():
        logits = qmodel(input_ids=ids).logits
    qmodel.save_pretrained(tmp_path, max_shard_size=200_000)
    files = sorted(os.listdir(tmp_path))
    assert "model.safetensors.index.json" in files and sum(f.endswith(".safetensors") for f in files) >= 3
    state = Q.load_state_dict_to_device(str(tmp_path))
    assert isinstance(state, Q.ShardedStateDict) and "model.layers.0.self_attn.q_proj.weight._data._data" in state
    again = Q.QuantizedModelForCausalLM.from_pretrained(tmp_path)
    with torch.no_grad():
        assert torch.equal(again(input_ids=ids).logits, logits)


@pytest.mark.gpu
def test_checkpoint_loads_straight_onto_the_device(tmp_path):
    """from_pretrained(device=cuda): every inner tensor of every quantized weight is created on the device by safetensors
    (no CPU staging) and the reloaded model reproduces the logits of the model that was saved."""
    model = tiny_llama(torch.bfloat16)
    qmodel = Q.QuantizedModelForCausalLM.quantize(model, weights=Q.qint4, exclude="lm_head")
    model.to("cuda")
    ids = torch.randint(1, 319, (2, 8)).to("cuda")
    with torch.no_grad():
        logits = qmodel(input_ids=ids).logits
    qmodel.save_pretrained(tmp_path, max_shard_size=200_000)
    state = Q.load_state_dict_to_device(str(tmp_path), torch.device("cuda", 0))
    assert all(state[k].is_cuda for k in list(state.keys())[:8])
    again = Q.QuantizedModelForCausalLM.from_pretrained(tmp_path, device=torch.device("cuda", 0))
    w = again.model.layers[1].mlp.down_proj.weight
    assert isinstance(w, Q.WeightQBitsTensor) and w._data._data.is_cuda and w._scale.is_cuda
    for (name, a), (_, b) in zip(model.state_dict().items(), again.state_dict().items()):
        assert a.device == b.device and torch.equal(a, b), name  # the flattened QTensors came back bit for bit
    with torch.no_grad():
        got = again(input_ids=ids).logits
    # not torch.equal: the non-persistent rotary table is rebuilt in fp32 on reload, the original model's was cast to bf16
    cos = torch.nn.functional.cosine_similarity(got.flatten().float(), logits.flatten().float(), dim=0)
    assert cos > 0.9999, float(cos)


@pytest.mark.gpu
@pytest.mark.parametrize("rows", [1, 32, 512])
def test_llama3_8b_layer_at_full_size_every_qlinear_against_exact_math(rows):
    """BASELINE config 5 at REAL size, one decoder layer (hidden 4096, intermediate 14336, 32 / 8 heads; small vocabulary so that
    the embedding does not dominate): int4 weights, q/k/v and gate/up linked by fuse_decode_projections.  Forward hooks capture the
    input and output of all seven QLinears inside the running model; every output is gated against float64 math on that module's
    own integers / scales and its captured input (decode M = 1: fused GEMV launches; batched decode M = 32: one streaming launch per
    sibling group; prefill M = 512: the fused int4 GEMM), and the fused model's logits equal the unfused model's where the
    arithmetic is the same."""
    import numpy as np
    from transformers import LlamaConfig, LlamaForCausalLM

    import optimum_quanto_amd as Q
    from helpers import assert_close_to_exact, to_numpy
    from oracle import quanto_oracle as O

    cfg = LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=1, num_attention_heads=32, num_key_value_heads=8,
                      vocab_size=512, max_position_embeddings=1024, rope_theta=500000.0, tie_word_embeddings=False)
    torch.manual_seed(0)
    with torch.device("cuda"):
        model = LlamaForCausalLM(cfg).to(torch.bfloat16).eval()
    Q.QuantizedModelForCausalLM.quantize(model, weights=Q.qint4, exclude="lm_head")
    ids = torch.randint(1, cfg.vocab_size - 1, (1, rows), generator=torch.Generator().manual_seed(rows)).cuda()
    with torch.no_grad():
        ref_logits = model(ids).logits
    assert Q.fuse_decode_projections(model) == 2
    captured = {}

    from optimum_quanto_amd.library.hip import quanto_hip

    def hook(name):
        def fn(mod, args, out):
            captured[name] = (args[0].detach().reshape(-1, args[0].shape[-1]), out.detach().reshape(-1, out.shape[-1]), quanto_hip.lib.last_kernel())
        return fn

    layer = model.model.layers[0]
    mods = {"q_proj": layer.self_attn.q_proj, "k_proj": layer.self_attn.k_proj, "v_proj": layer.self_attn.v_proj, "o_proj": layer.self_attn.o_proj,
            "gate_proj": layer.mlp.gate_proj, "up_proj": layer.mlp.up_proj, "down_proj": layer.mlp.down_proj}
    handles = [m.register_forward_hook(hook(n)) for n, m in mods.items()]
    with torch.no_grad():
        logits = model(ids).logits
    for h in handles:
        h.remove()
    assert set(captured) == set(mods)
    for name, m in mods.items():
        x, y, kernel = captured[name]
        assert x.shape[0] == rows
        w = m.weight
        N, K = w.shape
        packed, scale, shift = w._data._data.cpu().numpy(), to_numpy(w._scale), to_numpy(w._shift)
        if kernel == "dequant_mfma":
            # the large-M path does what the reference does: it multiplies the weight ROUNDED to bf16 (bit-identical to dequantize())
            wr = O.dequantize_qbits_ref(packed, 4, scale, shift, 0, 128, (N, K), "bf16").astype(np.float64)
            exact = np.matmul(to_numpy(x).astype(np.float64), wr.T)
        else:
            exact = O.qbits_mm_exact(to_numpy(x), packed, 4, scale, shift, 128, N, K)
        assert_close_to_exact(to_numpy(y), exact, "bf16", f"Llama-3-8B layer, {name} ({rows} rows, {K}->{N}, {kernel})")
    kernels = {n: captured[n][2] for n in mods}
    if rows == 1:
        assert kernels["q_proj"] == "gemv_multi" and kernels["gate_proj"] == "gemv_multi" and kernels["down_proj"] == "gemv", kernels
    elif rows == 32:
        assert kernels["q_proj"] == "skinny_multi" and kernels["down_proj"] == "skinny", kernels
    else:
        assert kernels["q_proj"] == "mfma_fused4" and kernels["down_proj"] == "mfma_fused4", kernels
    if rows <= 4:  # GEMV launches: the fused launch is bit-identical to the separate calls, so the logits are too
        assert torch.equal(logits, ref_logits)
    else:
        torch.testing.assert_close(logits.float(), ref_logits.float(), rtol=3e-2, atol=3e-2)
