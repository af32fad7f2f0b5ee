"""Deterministic weights / inputs shared by tools/gen_golden.py (build container, imports the
reference) and the tests (everywhere).  numpy RandomState (legacy MT19937 stream) so the values
regenerate bit-identically on any numpy version."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def cfg_dict(hidden=768, heads=12, layers=12, inter=None, vocab=30567, max_pos=512, img_dim=2054,
             eps=1e-12, **kw):
    d = dict(hidden_size=hidden, num_attention_heads=heads, num_hidden_layers=layers,
             intermediate_size=inter or 4 * hidden, vocab_size=vocab, max_position_embeddings=max_pos,
             type_vocab_size=2, img_feature_dim=img_dim, layer_norm_eps=eps, img_layer_norm_eps=eps,
             use_img_layernorm=1, add_residual=False, add_local_residual=False)
    d.update(kw)
    return d


def _lin(rs, sd, name, out_f, in_f, gain=1.4):
    sd[name + ".weight"] = (rs.standard_normal((out_f, in_f)) * (gain / np.sqrt(in_f))).astype(np.float32)
    sd[name + ".bias"] = (rs.standard_normal((out_f,)) * 0.1).astype(np.float32)


def _ln(rs, sd, name, h):
    sd[name + ".weight"] = (1.0 + 0.1 * rs.standard_normal((h,))).astype(np.float32)
    sd[name + ".bias"] = (0.1 * rs.standard_normal((h,))).astype(np.float32)


def layer_weights(rs, sd, p, h, inter, gain=1.4):
    for nm in ("query", "key", "value"):
        _lin(rs, sd, p + "attention.self." + nm, h, h, gain)
    _lin(rs, sd, p + "attention.output.dense", h, h, gain)
    _ln(rs, sd, p + "attention.output.LayerNorm", h)
    _lin(rs, sd, p + "intermediate.dense", inter, h, gain)
    _lin(rs, sd, p + "output.dense", h, inter, gain)
    _ln(rs, sd, p + "output.LayerNorm", h)


def bert_img_weights(rs, cfg, prefix="", seq=False, gain=1.4):
    """State dict with the key names of BertImgModel / SeqBertImgModel.  `gain` scales
    the encoder-layer weights: 1.4 gives deliberately sharp attention (scores up to +-10) for the
    kernel/encoder fixtures; the end-to-end logits fixture (G8) uses 1.0, closer to a trained model."""
    h, inter = cfg["hidden_size"], cfg["intermediate_size"]
    sd = {}
    e = prefix + "embeddings."
    sd[e + "word_embeddings.weight"] = (rs.standard_normal((cfg["vocab_size"], h)) * 0.5).astype(np.float32)
    sd[e + "word_embeddings.weight"][0] = 0
    sd[e + "position_embeddings.weight"] = (rs.standard_normal((cfg["max_position_embeddings"], h)) * 0.5).astype(np.float32)
    sd[e + "token_type_embeddings.weight"] = (rs.standard_normal((cfg["type_vocab_size"], h)) * 0.5).astype(np.float32)
    _ln(rs, sd, e + "LayerNorm", h)
    for i in range(cfg["num_hidden_layers"]):
        layer_weights(rs, sd, prefix + "encoder.layer.%d." % i, h, inter, gain)
    _lin(rs, sd, prefix + "pooler.dense", h, h)
    _lin(rs, sd, prefix + "img_embedding", h, cfg["img_feature_dim"])
    _ln(rs, sd, prefix + "LayerNorm", h)
    if seq:
        sd[prefix + "edge_dense.weight"] = (rs.standard_normal((1, h)) * 0.02).astype(np.float32)
    return sd


def cls_layer_lyx_weights(rs, sd, p, h, inter):
    layer_weights(rs, sd, p, h, inter)          # BertLayer members (attention.* unused by forward)
    _lin(rs, sd, p + "ensemble", 1, 2 * h)
    for nm in ("k_proj", "v_proj", "q_proj", "out_proj"):
        _lin(rs, sd, p + "cross_attention." + nm, h, h)
    _lin(rs, sd, p + "dense", h, h)
    _ln(rs, sd, p + "LayerNorm", h)


def calec_weights(rs, cfg, prefix="calec.", gain=1.4):
    """ChunkAlign_CLS_enc4_align_ensemble state dict (v10:872-889)."""
    h, inter = cfg["hidden_size"], cfg["intermediate_size"]
    sd = {}
    sd.update(bert_img_weights(rs, cfg, prefix + "global_enc.", gain=gain))
    sd.update(bert_img_weights(rs, cfg, prefix + "seq_enc.", seq=True, gain=gain))
    _lin(rs, sd, prefix + "cls_ensemble_1", h, 2 * h)
    for i in range(2):
        p = prefix + "cls_layer.%d." % i            # ClsLayer2: constructed, never called
        layer_weights(rs, sd, p, h, inter)
        for nm in ("cls_q_proj", "align_k_proj", "dense"):
            _lin(rs, sd, p + nm, h, h)
        _ln(rs, sd, p + "LayerNorm", h)
    for i in range(2):
        cls_layer_lyx_weights(rs, sd, prefix + "cls_layer_lyx.%d." % i, h, inter)
    _lin(rs, sd, prefix + "classifier", 2, h)
    _lin(rs, sd, prefix + "fusion_align", 1024, 2 * h)
    _lin(rs, sd, prefix + "prior", 1, h)
    return sd


def enc4_align_weights(rs, cfg, prefix="", gain=1.0):
    """ChunkAlign_CLS_enc4_align state dict (v10:1016-1027): both encoders, cls_ensemble, three ClsLayer2, classifier."""
    h, inter = cfg["hidden_size"], cfg["intermediate_size"]
    sd = {}
    sd.update(bert_img_weights(rs, cfg, prefix + "global_enc.", gain=gain))
    sd.update(bert_img_weights(rs, cfg, prefix + "seq_enc.", seq=True, gain=gain))
    _lin(rs, sd, prefix + "cls_ensemble", h, 2 * h)
    for i in range(3):
        p = prefix + "cls_layer.%d." % i
        layer_weights(rs, sd, p, h, inter)          # BertLayer members (attention.* unused by ClsLayer2.forward)
        for nm in ("cls_q_proj", "align_k_proj", "dense"):
            _lin(rs, sd, p + nm, h, h)
        _ln(rs, sd, p + "LayerNorm", h)
    _lin(rs, sd, prefix + "classifier", 2, h)
    return sd


def abstract_specific_weights(rs, cfg):
    """Abstract_Specific state dict minus roberta.* (modeling_ensemble.py:425-458)."""
    sd = calec_weights(rs, cfg, "calec.", gain=1.0)
    _lin(rs, sd, "classifier", 1, 768 + 768)
    _lin(rs, sd, "abst_confidence_scorer", 1, 1024)
    _lin(rs, sd, "confidence_scorer", 1, 768)
    for nm in ("mapping_network_alignment", "mapping_network_vision"):
        _lin(rs, sd, nm + ".1", 768 * 5, 768)
        _lin(rs, sd, nm + ".4", 1024 * 5, 768 * 5)
    sd["promptfuse.weight"] = (rs.standard_normal((2, 1024)) * 0.02).astype(np.float32)
    return sd


def to_torch(sd, dtype=torch.float32):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in sd.items()}


def stub_roberta_pooled(prefix_emb, r_ids):
    """Deterministic stand-in for the missing RoBERTa body: a fixed function of the prefix and ids
    (keeps the head/loss/gradient plumbing testable; SURVEY 8c 'parity unpinned' boundary)."""
    n = prefix_emb.shape[0]
    base = torch.tanh(prefix_emb.mean(dim=1))
    bump = torch.sin(r_ids.to(prefix_emb.dtype).sum(dim=1, keepdim=True) * 1e-3
                     + torch.arange(1024, dtype=prefix_emb.dtype)[None, :] * 0.01)
    return base + 0.1 * bump


def collate_samples(seed=3, n_examples=3, n_regions=10, img_dim=14):
    """Ragged per-choice 19-tuples in the layout the reference's datasets return (Data/VCRChunkAlign.py:596-688):
    (img_id, image, text, r_input_ids, r_segment_ids, r_input_mask, input_ids, segment_ids, input_mask, img_feat, img_mask,
    target, chunk_mask, gather_index, offsets, ques, ans, total_label, align_pos).  Shared by tools/gen_golden.py (G12: fed to
    the REFERENCE's SNLIGPT_gen_collate) and the collate tests."""
    rs = np.random.RandomState(seed)
    examples = []
    for e in range(n_examples):
        choices = []
        nreg = int(rs.randint(3, n_regions - 1))
        img_feat = torch.from_numpy(rs.standard_normal((n_regions, img_dim)).astype(np.float32))
        img_mask = torch.cat([torch.ones(nreg), torch.zeros(n_regions - nreg)])
        for c in range(4):
            ln = int(rs.randint(5, 12))
            rl = int(rs.randint(6, 15))
            offs, k = [], 1
            while k < ln - 1:
                w = min(int(rs.randint(1, 4)), ln - 1 - k)
                offs.append(list(range(k, k + w)))
                k += w
            gi = torch.tensor([i for i, o in enumerate(offs) for _ in o], dtype=torch.int64)
            cm = torch.from_numpy((rs.uniform(size=(ln, ln)) < 0.5).astype(np.float32))
            tl = torch.from_numpy(rs.randint(0, 3, size=ln).astype(np.int64))
            choices.append(("id%d" % e, "", "", torch.from_numpy(rs.randint(3, 99, size=rl)), torch.zeros(rl, dtype=torch.int64),
                            torch.ones(rl), torch.from_numpy(rs.randint(3, 99, size=ln)), torch.ones(ln, dtype=torch.int64),
                            torch.ones(ln), img_feat, img_mask, torch.tensor(int(c == e % 4)), cm, gi, offs, "q%d" % e, "a%d" % c, tl,
                            (tl != 0).to(torch.int64)))
        examples.append(tuple(choices))
    return examples


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))


def report_use(what, used, tol, kind="max|err|/scale"):
    """MODCR_TEST_REPORT=1 (pytest -s): one line per tolerance check -- test id, quantity, error, bound, share of the bound used.
    profiles/r04_tolerance_report.txt is this output; the bounds in the tests are set from it (<= ~2x the observed use)."""
    if os.environ.get("MODCR_TEST_REPORT"):
        test = os.environ.get("PYTEST_CURRENT_TEST", "").split(" ")[0].split("::", 1)[-1]
        print("  [tol] %-70s %-38s %s %.3e  bound %.1e  used %3.0f %%" % (test[:70], str(what)[:38], kind, used, tol, 100.0 * used / tol if tol else 0.0))


# ---- attention-probability dropout mask: host restatement of csrc/attn_common.h (round-5 layout) -----------------------------
def _fmix32(x):
    x &= 0xffffffff
    x ^= x >> 16; x = x * 0x85EBCA6B & 0xffffffff; x ^= x >> 13; x = x * 0xC2B2AE35 & 0xffffffff; x ^= x >> 16
    return x


def attn_drop_const(j):
    """multiplier of word j: fmix32((j + 1) * 0x9E3779B1) | 1"""
    return _fmix32((j + 1) * 0x9E3779B1) | 1


def attn_drop_keep_torch(seq_ids, heads, s, p, seed, offset, device="cpu", keys=None):
    """keep[i, head, query, key] (float 0 / 1) for sequences `seq_ids`, queries 0..s-1, keys 0..keys-1 (default s):
    base(row, l4) = fold(cm * 0x85EBCA6B + K), cm = (row * 4 + l4) * 0x9E3779B1 mod 2^32, row = (n * A + head) * 256 + query;
    word j = fold(base * C[j] + K), fold = low ^ high half of the 64-bit sum; key -> l4 = (key >> 2) & 3,
    j = 2 (key >> 4) + ((key >> 1) & 1), 16-bit field key & 1; kept iff the field as a signed 16-bit number is
    >= round(p * 2^16) - 32768.  int64 arithmetic wraps mod 2^64, which is what the 64-bit multiply-add does."""
    m32 = 0xffffffff
    keys = s if keys is None else keys
    key64 = (seed + offset * 0x9E3779B97F4A7C15) & (2 ** 64 - 1)
    k = key64 - 2 ** 64 if key64 >= 2 ** 63 else key64
    thr = min(max(int(p * 65536 + 0.5), 1), 65535) - 32768

    def fold(a, c):
        c = c - 2 ** 64 if c >= 2 ** 63 else c
        prod = a * c + k
        return (prod & m32) ^ ((prod >> 32) & m32)

    n_ = torch.as_tensor(list(seq_ids), dtype=torch.int64, device=device).view(-1, 1, 1, 1)
    a_ = torch.arange(heads, dtype=torch.int64, device=device).view(1, -1, 1, 1)
    q_ = torch.arange(s, dtype=torch.int64, device=device).view(1, 1, -1, 1)
    l4_ = torch.arange(4, dtype=torch.int64, device=device).view(1, 1, 1, -1)
    row = (n_ * heads + a_) * 256 + q_
    cm = ((row * 4 + l4_) * 0x9E3779B1) & m32
    base = fold(cm, 0x85EBCA6B)                                      # [n, heads, s, 4]
    key_ = torch.arange(keys, dtype=torch.int64, device=device)
    jj = 2 * (key_ >> 4) + ((key_ >> 1) & 1)
    cs = torch.tensor([attn_drop_const(int(j)) for j in jj.tolist()], dtype=torch.int64, device=device)   # [keys]
    b = base[..., (key_ >> 2) & 3]                                   # [n, heads, s, keys]
    prod = b * cs + k
    w = (prod & m32) ^ ((prod >> 32) & m32)
    f = (w >> ((key_ & 1) * 16)) & 0xffff
    f = torch.where(f >= 32768, f - 65536, f)
    return (f >= thr).to(torch.float32)
