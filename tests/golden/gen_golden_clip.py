#!/usr/bin/env python
"""Golden vectors for the CLIP text tower, generated in the build container.

The reference's FrozenCLIPEmbedder cannot be constructed offline (encoders/modules.py:184-185: from_pretrained), and
its arithmetic is the third-party `transformers` CLIPTextModel anyway.  This script builds a RANDOMLY INITIALISED
transformers.CLIPTextModel from a config (no download), loads seeded synthetic weights into it and drives its OWN
sub-modules the way the reference's patched forwards do (encoders/modules.py:198-371):
    embeddings(inputs_embeds = token_embedding(ids) [patched by the EmbeddingManager])  ->  + position embeddings
    encoder(..., causal mask, output_hidden_states)                                     ->  every layer's input + the output
    final_layer_norm(0.5 * states[-2] + 0.5 * states[-1])
Only inputs and outputs are stored (tests/golden/golden_clip.npz); weights come from the oracle's seeded generator.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden_clip.py
"""
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import clip_oracle as CO            # noqa: E402
from oracle import ldm_oracle as O              # noqa: E402
from transformers import CLIPTextConfig, CLIPTextModel           # noqa: E402
from transformers.masking_utils import create_causal_mask        # noqa: E402


def hf_model(cfg: CO.ClipConfig, sd):
    hc = CLIPTextConfig(vocab_size=cfg.vocab, hidden_size=cfg.hidden, intermediate_size=cfg.intermediate,
                        num_hidden_layers=cfg.layers, num_attention_heads=cfg.heads, max_position_embeddings=cfg.max_pos,
                        hidden_act="quick_gelu", layer_norm_eps=cfg.eps, bos_token_id=0, eos_token_id=1, pad_token_id=1)
    m = CLIPTextModel(hc).eval()
    own = {k[len(CO.CLIP_PREFIX):]: v for k, v in sd.items()}
    tgt = m.state_dict()
    # transformers >= 5 dropped the `text_model.` level; older versions keep it
    mapped = {}
    for k in tgt:
        kk = k[len("text_model."):] if k.startswith("text_model.") else k
        if kk in own:
            mapped[k] = own[kk]
    missing = [k for k in tgt if k not in mapped and "position_ids" not in k]
    assert not missing, missing
    m.load_state_dict(mapped, strict=False)
    return m


@torch.no_grad()
def hf_forward(m, inputs_embeds, skip_weights=(0.5, 0.5)):
    tm = getattr(m, "text_model", m)
    hs = tm.embeddings(inputs_embeds=inputs_embeds)
    mask = create_causal_mask(config=m.config, inputs_embeds=hs, attention_mask=None, past_key_values=None)
    # every layer's INPUT through forward pre-hooks (transformers >= 5 records hidden states only at the top-level call)
    states, hooks = [], []
    for layer in tm.encoder.layers:
        hooks.append(layer.register_forward_pre_hook(lambda mod, args, kwargs: states.append(args[0] if args else kwargs["hidden_states"]),
                                                     with_kwargs=True))
    enc = tm.encoder(inputs_embeds=hs, attention_mask=mask, is_causal=True)
    for hk in hooks:
        hk.remove()
    states.append(enc.last_hidden_state)
    assert len(states) == m.config.num_hidden_layers + 1
    w = torch.tensor(skip_weights) / sum(skip_weights)
    x = sum(wi * st for wi, st in zip(w, states[-len(w):]))
    return tm.final_layer_norm(x), states


def main():
    out = {}
    g = torch.Generator().manual_seed(4)
    # ---- tiny tower: plain prompt batch, and a batch patched by the EmbeddingManager restatement ----
    cfg = CO.TINY_CLIP
    sd = O.synth_state_dict(CO.clip_param_shapes(cfg), seed=41)
    m = hf_model(cfg, sd)
    ids = torch.randint(2, cfg.vocab, (3, 77), generator=g)
    ids[:, 0] = 0
    ids[0, 20:] = 1; ids[1, 9:] = 1; ids[2, 70:] = 1
    emb = CO.clip_embed_tokens(sd, ids)
    z, states = hf_forward(m, emb)
    out["tiny_ids"] = ids.numpy()
    out["tiny_z"] = z.numpy()
    out["tiny_state_last_in"] = states[-2].numpy()
    z2, _ = hf_forward(m, emb, skip_weights=(0.2, 0.8))
    out["tiny_z_w28"] = z2.numpy()
    token, K = 777, 4
    ids_p = ids.clone()
    ids_p[0, 5] = token; ids_p[2, 11] = token; ids_p[2, 30] = token      # second occurrence in row 2 must be ignored
    subj = torch.randn(16, K, cfg.hidden, generator=g) * 0.07
    patched, _, _ = CO.embedding_manager_patch(ids_p, CO.clip_embed_tokens(sd, ids_p), token, subj)
    zp, _ = hf_forward(m, patched)
    out["tiny_ids_subj"] = ids_p.numpy()
    out["tiny_subj_emb"] = subj.numpy()
    out["tiny_z_subj"] = zp.numpy()                                     # [48, 77, 64]
    # ---- full-size tower (openai/clip-vit-large-patch14 shape), two prompts ----
    cfg = CO.SD15_CLIP
    sd = O.synth_state_dict(CO.clip_param_shapes(cfg), seed=42)
    m = hf_model(cfg, sd)
    ids = torch.randint(2, cfg.vocab, (2, 77), generator=g)
    ids[:, 0] = 0
    ids[0, 12:] = 1; ids[1, 40:] = 1
    z, _ = hf_forward(m, CO.clip_embed_tokens(sd, ids))
    out["sd15_ids"] = ids.numpy()
    out["sd15_z"] = z.numpy().astype(np.float32)
    # ---- zero-shot identity path on tiny towers: two CLIPTextModel instances driven the way
    # arc2face_forward_face_embs (ldm/util.py:1085-1131) and arc2face_inverse_face_prompt_embs (:1138-1233, called by
    # SubjBasisGenerator.forward, subj_basis_generator.py:482-560) drive CLIPTextModelWrapper (arc2face_models.py:175-280):
    # plain last hidden state for the first, the last THREE states weighted [1, 2, 4] / 7 for the second.  The prompts'
    # token ids are stand-ins (no tokenizer offline): position 4 holds the 'id' token / positions 4..19 the placeholders.
    cfg = CO.TINY_CLIP
    g2 = torch.Generator().manual_seed(9)
    sd_a = O.synth_state_dict(CO.clip_param_shapes(cfg), seed=43)      # "arc2face text encoder"
    sd_p = O.synth_state_dict(CO.clip_param_shapes(cfg), seed=44)      # "prompt2token_proj"
    m_a, m_p = hf_model(cfg, sd_a), hf_model(cfg, sd_p)
    sd_t = O.synth_state_dict(CO.clip_param_shapes(cfg), seed=41)
    m_t = hf_model(cfg, sd_t)
    emb_t = CO.clip_embed_tokens(sd_t, torch.tensor(out["tiny_ids"]))
    out["tiny_z_w1"] = hf_forward(m_t, emb_t, skip_weights=(1.0,))[0].numpy()
    out["tiny_z_w124"] = hf_forward(m_t, emb_t, skip_weights=(1.0, 2.0, 4.0))[0].numpy()
    id_dim, id_token, pad_token = 48, 333, 1
    face = torch.randn(2, id_dim, generator=g2)
    face = face / face.norm(dim=1, keepdim=True)
    ids_a = torch.tensor([[0, 11, 12, 13, id_token, 14] + [pad_token] * 71])                 # "photo of a id person"
    ids_a = ids_a.repeat(2, 1)
    tok = CO.clip_embed_tokens(sd_a, ids_a).clone()
    tok[ids_a == id_token] = torch.nn.functional.pad(face, (0, cfg.hidden - id_dim))
    full_a, _ = hf_forward(m_a, tok, skip_weights=(1.0,))
    core_a = full_a[:, 4:20]
    ids_p = torch.tensor([[0, 11, 12, 13] + [15] * 16 + [pad_token] * 57]).repeat(2, 1)      # "photo of a , , ... ,"
    tokp = CO.clip_embed_tokens(sd_p, ids_p).clone()
    tokp[:, 4:20] = core_a
    full_p, _ = hf_forward(m_p, tokp, skip_weights=(1.0, 2.0, 4.0))
    out["zs_face"] = face.numpy()
    out["zs_ids_arc2face"] = ids_a.numpy()
    out["zs_ids_inverse"] = ids_p.numpy()
    out["zs_arc2face_full"] = full_a.numpy()
    out["zs_inverse_full"] = full_p.numpy()
    np.savez_compressed(ROOT / "tests" / "golden" / "golden_clip.npz", **out)
    for k, v in out.items():
        print(k, v.shape, v.dtype)


if __name__ == "__main__":
    main()
