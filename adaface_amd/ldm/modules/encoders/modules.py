"""Drop-in for ldm.modules.encoders.modules.FrozenCLIPEmbedder (reference encoders/modules.py:177-463): the CLIP text
tower that turns prompts into the [B*16, 77, 768] static prompt embedding the denoising path consumes — the step in
FRONT of the path.  Same constructor kwargs and call surface (`forward(text, embedding_manager=...)`,
`encode`, `set_last_layers_skip_weights`, `sample_last_layers_skip_weights`, `freeze`), same state_dict keys
(`transformer.text_model.*`, so an SD checkpoint's `cond_stage_model.*` entries load unchanged).

The reference wraps transformers.CLIPTextModel and patches four of its forwards so that an EmbeddingManager can rewrite
the token embeddings between the lookup and the encoder (modules.py:198-227) and so that the last hidden states are
blended before the final LayerNorm (modules.py:361-370).  Here the same sequence is three calls:
    af_clip_embed_tokens  ->  embedding_manager(input_ids, inputs_embeds)  ->  af_clip_text_forward
and the arithmetic runs on the HIP kernels of the denoising path (LayerNorm, GEMM, causal attention).

The tokenizer is NOT part of this package: `CLIPTokenizer.from_pretrained(version)` needs vocabulary files that do not
exist offline.  It is looked up lazily, only when `forward` is given strings; token-id tensors are accepted directly.
"""
from __future__ import annotations

import numpy as np
import torch

from adaface_amd import layout
from adaface_amd.ldm._hipmodule import HipModule, build_param_tree

# openai/clip-vit-large-patch14, text tower
CLIP_VIT_L14_TEXT = dict(vocab=49408, hidden=768, layers=12, heads=12, intermediate=3072, max_pos=77)


class FrozenCLIPEmbedder(HipModule):
    _ckpt_prefix = "cond_stage_model."

    def __init__(self, version="openai/clip-vit-large-patch14", device="cuda", max_length=77,
                 last_layers_skip_weights=[0.5, 0.5], randomize_clip_skip_weights=False, clip_config=None,
                 tokenizer=None):
        super().__init__()
        self.version = version
        self.device = device
        self.max_length = max_length
        self.clip_config = dict(CLIP_VIT_L14_TEXT if clip_config is None else clip_config)
        if self.clip_config["max_pos"] < max_length:
            raise ValueError("max_length exceeds the tower's position table")
        self.tokenizer = tokenizer            # resolved lazily in tokenize()
        shapes = {"transformer.text_model." + k: v for k, v in layout.clip_text_param_shapes(**self.clip_config).items()}
        build_param_tree(self, shapes)
        self.set_last_layers_skip_weights(last_layers_skip_weights, use_as_dirichlet_weights=randomize_clip_skip_weights)

    def _engine_kwargs(self):
        return {"clip": dict(self.clip_config)}

    # ---- modules.py:397-427 ----
    def set_last_layers_skip_weights(self, weights, use_as_dirichlet_weights=False):
        if not use_as_dirichlet_weights:
            w = np.array(weights, dtype=np.float64)
            self.last_layers_skip_weights = w / np.sum(w)
            self.dir_sampler = None
        else:
            self.dir_sampler = torch.distributions.dirichlet.Dirichlet(torch.tensor(weights, dtype=float))
            self.sample_last_layers_skip_weights()

    def sample_last_layers_skip_weights(self, verbose=False):
        if self.dir_sampler is None:
            return
        self.last_layers_skip_weights = self.dir_sampler.sample().numpy()

    def freeze(self):
        self.eval()
        for p in self.parameters():
            p.requires_grad = False

    # ---- tokenizer (modules.py:441-444) ----
    def tokenize(self, text):
        if self.tokenizer is None:
            why, tok = None, None
            try:
                from transformers import CLIPTokenizer
                tok = CLIPTokenizer.from_pretrained(self.version)
                # transformers >= 5 does not raise when the vocabulary files are missing: it returns a tokenizer with a
                # two-entry vocabulary that maps every word to the same id.  Refuse it instead of encoding garbage.
                if len(tok) != self.clip_config["vocab"]:
                    why = f"returned a tokenizer with {len(tok)} entries, the tower's table has {self.clip_config['vocab']}"
            except Exception as e:  # no vocabulary files offline
                why = f"raised {type(e).__name__}"
            if why is not None:
                raise RuntimeError(
                    f"CLIPTokenizer.from_pretrained('{self.version}') {why}: the vocabulary files are not available; "
                    "pass `tokenizer=` (any callable with the CLIPTokenizer call signature) or feed token ids")
            self.tokenizer = tok
        enc = self.tokenizer(text, truncation=True, max_length=self.max_length, return_length=True,
                             return_overflowing_tokens=False, padding="max_length", return_tensors="pt")
        return enc["input_ids"]

    @torch.no_grad()
    def forward(self, text, embedding_manager=None, **kwargs):
        """text: list of prompts, or an int64 tensor [B, max_length] of token ids.  Returns [B(*16), T, hidden] fp32."""
        if kwargs:
            raise TypeError(f"FrozenCLIPEmbedder.forward: unexpected arguments {sorted(kwargs)}")
        tokens = text if isinstance(text, torch.Tensor) else self.tokenize(text)
        dev = torch.device(self.device) if not isinstance(self.device, torch.device) else self.device
        if dev.type != "cuda":
            raise RuntimeError("FrozenCLIPEmbedder runs on a HIP device only (adaface_amd has no CPU path)")
        tokens = tokens.to(dev).long().contiguous()
        eng = self.engine(dev)
        inputs_embeds = eng.clip_embed_tokens(tokens)                         # modules.py:207-208
        if embedding_manager is not None:                                     # modules.py:214-215
            inputs_embeds = embedding_manager(tokens, inputs_embeds)
        w = np.asarray(self.last_layers_skip_weights, dtype=np.float64)
        if w.shape != (2,):
            raise NotImplementedError("the last-layers blend is built for two weights (the reference's setting)")
        return eng.clip_text_forward(inputs_embeds, float(w[0]), float(w[1]))

    def encode(self, text, **kwargs):
        return self(text, **kwargs)
