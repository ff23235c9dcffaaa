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
