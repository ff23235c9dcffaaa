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
