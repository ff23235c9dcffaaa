#!/usr/bin/env python3
"""BASELINE config 5: Llama-3-8B (random init, bf16), weights=qint4 with lm_head excluded, decode tokens/s at batch 1 / 32.

    python scripts/bench_generate.py [--layers 32] [--batch 1 32] [--prompt 512] [--new 128] [--weights qint4]

Three drivers:
  * "reference": the reference's own method, bench/generation/metrics/latency.py:24-94, call for call - ``model.generate`` with
    ``GenerationConfig(max_new_tokens = min_new_tokens = 512, use_cache, num_beams=1, do_sample=False, eos_token_id=None)`` on a
    random prompt of 512 tokens and an all-ones attention mask, device events around the whole call, mean over ``--iterations``
    calls divided by the number of new tokens (so the prefill is inside the figure, as in the reference);
  * "eager": one Python-issued forward per token on a static KV cache (decode only; host-bound at batch 1);
  * "graph": the single-token forward (static KV cache) captured once in a hipGraph and replayed per token (decode only).
``--fuse`` links sibling projections (q/k/v, gate/up) so that a decode step issues ONE ``quanto::qbits_mm_multi`` launch for them
(optimum_quanto_amd.fuse_decode_projections).  Prints one JSON line per (batch, driver).  No network: the model is created from a config.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_model(layers, weights, device):
    from transformers import LlamaConfig, LlamaForCausalLM

    import optimum_quanto_amd as Q

    cfg = LlamaConfig(hidden_size=4096, intermediate_size=14336, num_hidden_layers=layers, num_attention_heads=32,
                      num_key_value_heads=8, vocab_size=128256, max_position_embeddings=8192, rope_theta=500000.0,
                      tie_word_embeddings=False)
    torch.manual_seed(0)
    t0 = time.time()
    with torch.device(device):
        model = LlamaForCausalLM(cfg).to(torch.bfloat16).eval()
    t1 = time.time()
    if weights != "none":
        Q.QuantizedModelForCausalLM.quantize(model, weights=weights, exclude="lm_head")
    torch.cuda.synchronize()
    print(f"# model built in {t1 - t0:.1f}s, quantized in {time.time() - t1:.1f}s, "
          f"{torch.cuda.memory_allocated() / 2**30:.1f} GiB allocated", file=sys.stderr)
    return model, cfg


@torch.no_grad()
def run_reference_method(model, cfg, batch, prompt, new, iterations, device):
    """bench/generation/metrics/latency.py:24-94 (tokenizer-free: pad_token_id only matters with an eos token, which is disabled)."""
    from transformers import GenerationConfig

    gen = GenerationConfig(max_new_tokens=new, min_new_tokens=new, use_cache=True, pad_token_id=0, num_beams=1, do_sample=False,
                           eos_token_id=None)
    if getattr(model, "generation_config", None) is not None:
        model.generation_config.eos_token_id = None
    torch.cuda.synchronize()
    ids = torch.randint(1, cfg.vocab_size - 1, size=(batch, prompt)).to(device)
    masks = torch.ones(batch, prompt, dtype=torch.int32).to(device)
    lat = []
    for _ in range(iterations):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        out = model.generate(ids, attention_mask=masks, generation_config=gen)
        e1.record()
        torch.cuda.synchronize()
        assert out.shape[1] == prompt + new
        lat.append(e0.elapsed_time(e1))
    return lat


@torch.no_grad()
def run(model, cfg, batch, prompt, new, driver, device):
    from transformers import StaticCache

    ids = torch.randint(1, cfg.vocab_size - 1, (batch, prompt), device=device)
    cache = StaticCache(config=cfg, max_batch_size=batch, max_cache_len=prompt + new + 8, device=device, dtype=torch.bfloat16)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    ev[0].record()
    out = model(input_ids=ids, past_key_values=cache, cache_position=torch.arange(prompt, device=device), use_cache=True)
    tok = out.logits[:, -1:].argmax(-1)
    ev[1].record()
    pos = torch.tensor([prompt], device=device)

    def step(tok, pos):
        o = model(input_ids=tok, past_key_values=cache, cache_position=pos, use_cache=True)
        return o.logits[:, -1:].argmax(-1)

    if driver == "eager":
        for _ in range(new):
            tok = step(tok, pos)
            pos += 1
    else:
        s_tok, s_pos = tok.clone(), pos.clone()
        for _ in range(2):  # warm-up on the side stream before capture
            step(s_tok, s_pos)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            s_out = step(s_tok, s_pos)
        torch.cuda.synchronize()
        ev[1].record()
        for _ in range(new):
            g.replay()
            s_tok.copy_(s_out)
            s_pos += 1
    ev[2].record()
    torch.cuda.synchronize()
    prefill_ms, decode_ms = ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2])
    return prefill_ms, decode_ms


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--batch", type=int, nargs="+", default=[1, 32])
    ap.add_argument("--prompt", type=int, default=512)
    ap.add_argument("--new", type=int, default=512)
    ap.add_argument("--weights", default="qint4")
    ap.add_argument("--drivers", nargs="+", default=["reference", "graph"])
    ap.add_argument("--iterations", type=int, default=3, help="generate() calls per batch size for the reference method (the reference uses 10)")
    ap.add_argument("--fuse", action="store_true", help="one launch for q/k/v and for gate/up at decode time")
    args = ap.parse_args()
    device = torch.device("cuda", 0)
    model, cfg = build_model(args.layers, args.weights, device)
    linked = 0
    if args.fuse:
        import optimum_quanto_amd as Q

        linked = Q.fuse_decode_projections(model)
    for b in args.batch:
        for d in args.drivers:
            try:
                if d == "reference":
                    lat = run_reference_method(model, cfg, b, args.prompt, args.new, args.iterations, device)
                    mean_ms = sum(lat) / len(lat)
                    print(json.dumps({"config": "Llama-3-8B random-init bf16", "layers": args.layers, "weights": args.weights, "batch": b,
                                      "prompt": args.prompt, "new_tokens": args.new, "driver": "reference method (model.generate, prefill included)",
                                      "fused_projection_groups": linked, "iterations": len(lat),
                                      "latency_per_token_ms": round(mean_ms / args.new, 3),
                                      "tokens_per_s": round(b * args.new / (mean_ms * 1e-3), 1)}), flush=True)
                    continue
                prefill_ms, decode_ms = run(model, cfg, b, args.prompt, args.new, d, device)
                print(json.dumps({"config": "Llama-3-8B random-init bf16", "layers": args.layers, "weights": args.weights,
                                  "batch": b, "prompt": args.prompt, "new_tokens": args.new, "driver": d, "fused_projection_groups": linked,
                                  "prefill_ms": round(prefill_ms, 2), "ms_per_token": round(decode_ms / args.new, 3),
                                  "decode_tokens_per_s": round(b * args.new / (decode_ms * 1e-3), 1)}), flush=True)
            except Exception as e:  # keep going: one driver failing must not hide the other's number
                print(json.dumps({"batch": b, "driver": d, "error": repr(e)[:300]}), flush=True)


if __name__ == "__main__":
    main()
