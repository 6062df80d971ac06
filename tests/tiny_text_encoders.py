"""Tiny seeded stand-ins for T5-XXL / CLIP-L and their tokenizers (tests and golden generation only).

Same classes as the real ones (transformers' T5EncoderModel / CLIPTextModel), random weights from a fixed seed, and a
toy tokenizer with the HF call signature (character codes as ids)."""
import types

import torch


class ToyTokenizer:
    def __init__(self, vocab=60, eos=61, pad=0, model_max_length=77):
        self.vocab, self.eos, self.pad, self.model_max_length = vocab, eos, pad, model_max_length

    def __call__(self, prompt, padding="max_length", max_length=None, truncation=True, return_length=False,
                 return_overflowing_tokens=False, return_tensors="pt"):
        prompt = [prompt] if isinstance(prompt, str) else prompt
        rows = [[1 + (ord(c) % (self.vocab - 1)) for c in p] + [self.eos] for p in prompt]
        if padding == "longest":
            max_length = max(len(r) for r in rows)
        if truncation and max_length is not None:
            rows = [r[:max_length - 1] + [self.eos] if len(r) > max_length else r for r in rows]
        rows = [r + [self.pad] * (max_length - len(r)) for r in rows]
        return types.SimpleNamespace(input_ids=torch.tensor(rows, dtype=torch.long))

    def batch_decode(self, ids):
        return ["".join(chr(96 + int(t) % 26) for t in row) for row in ids]


def build(seed=1234, dtype=torch.float32):
    from transformers import CLIPTextConfig, CLIPTextModel, T5Config, T5EncoderModel
    torch.manual_seed(seed)
    clip = CLIPTextModel(CLIPTextConfig(vocab_size=64, hidden_size=32, intermediate_size=64, num_hidden_layers=2,
                                        num_attention_heads=4, max_position_embeddings=77, projection_dim=32,
                                        eos_token_id=61, bos_token_id=62, pad_token_id=0)).eval().to(dtype)
    t5 = T5EncoderModel(T5Config(vocab_size=64, d_model=48, d_kv=8, d_ff=64, num_layers=2, num_heads=4)).eval().to(dtype)
    return [clip, t5], [ToyTokenizer(), ToyTokenizer(model_max_length=512)]
