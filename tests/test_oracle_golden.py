"""Pin the CPU oracle (oracle/modcr_oracle.py) against golden vectors produced by the reference's
own modules (tools/gen_golden.py).  Tolerance: restatement vs reference 1e-5 fp32;
a few deep 12-layer outputs get 5e-5 because op order differs in the last bit per layer."""
import numpy as np
import pytest
import torch

import helpers as H
from oracle import modcr_oracle as O


def t(a):
    return torch.from_numpy(np.asarray(a))


def close(a, b, tol=1e-5):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    b = np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    err = np.abs(a - b).max()
    scale = max(1.0, np.abs(b).max())
    assert err <= tol * scale, "max abs err %.3e (scale %.2f)" % (err, scale)


def gi_list(gi_pad):
    return [t(row[row >= 0]) for row in np.asarray(gi_pad)]


def test_g1_self_attention_with_and_without_history():
    g = H.load_golden("G1_self_attention")
    n, s, h, a, p = g["shape"]
    rs = np.random.RandomState(int(g["seed"]))
    sd = {}
    for nm in ("query", "key", "value"):
        H._lin(rs, sd, nm, h, h)
    sd = H.to_torch(sd)
    x, mask = t(g["x"]), t(g["mask"])
    ctx, probs = O.self_attention(x, O.extend_mask(mask), sd, "", a)
    close(ctx, g["ctx"]); close(probs, g["probs"])
    maskp = torch.cat([torch.ones(n, p), mask], dim=1)
    ctx, probs = O.self_attention(x, O.extend_mask(maskp), sd, "", a, history_state=t(g["hist"]))
    close(ctx, g["ctx_hist"]); close(probs, g["probs_hist"])


def test_g2_chunk_mean_query_attention():
    g = H.load_golden("G2_chunk_cross_attention")
    n, tt, r, h, a = g["shape"]
    rs = np.random.RandomState(int(g["seed"]))
    sd = {}
    for nm in ("query", "key", "value"):
        H._lin(rs, sd, nm, h, h)
    sd = H.to_torch(sd)
    ctx, probs = O.self_attention(t(g["x"]), O.extend_mask(t(g["mask"])), sd, "", a,
                                  gather_index=gi_list(g["gather_index"]))
    close(ctx, g["ctx"]); close(probs, g["probs"])


@pytest.mark.parametrize("name,full", [("G3_layer_h128", True), ("G3_layer_h768", False),
                                        ("G9_layer_h1024", False)])
def test_g3_layer_forward_backward(name, full):
    g = H.load_golden(name)
    n, s, h, a = g["shape"]
    rs = np.random.RandomState(int(g["seed"]))
    sd = {}
    H.layer_weights(rs, sd, "", h, 4 * h)
    sd = {k: v.requires_grad_(True) for k, v in H.to_torch(sd).items()}
    x = t(g["x"]).requires_grad_(True)
    y, _ = O.bert_layer(x, O.extend_mask(t(g["mask"])), sd, "", a, 1e-12)
    close(y, g["y"])
    (y * t(g["dy"])).sum().backward()
    close(x.grad, g["dx"], 2e-5)
    for k, v in sd.items():
        if full:
            close(v.grad, g["grad." + k], 2e-5)
        else:
            close(v.grad.reshape(-1)[:64], g["ghead." + k], 2e-5)
            ref = g["gsum." + k]
            assert abs(v.grad.abs().sum().item() - ref[1]) <= 1e-4 * max(1.0, ref[1])


def test_g4_phase_masks():
    g = H.load_golden("G4_phase_masks")
    im, cm = t(g["input_mask"]), t(g["chunk_attention_mask"])
    tt = cm.shape[1]
    r = im.shape[1] - tt
    for key, phase in (("phase1", 1), ("phase1_l2", 1), ("phase2", 2), ("phase2_l8", 2),
                       ("phase3", 3), ("phase3_l11", 3)):
        m = O.seq_phase_mask(im, cm, tt, r, phase)
        assert np.array_equal(m.numpy(), g[key]), key


def _small_cfg():
    return H.cfg_dict(hidden=128, heads=2, layers=12, vocab=30567, max_pos=64, img_dim=70)


def test_g5_encoders_end_to_end():
    g = H.load_golden("G5_encoders_small")
    cfg = _small_cfg()
    rs = np.random.RandomState(int(g["seed"]))
    sd_g = H.to_torch(H.bert_img_weights(rs, cfg))
    sd_s = H.to_torch(H.bert_img_weights(rs, cfg, seq=True))
    ids, tt, im, img = t(g["input_ids"]), t(g["token_type_ids"]), t(g["input_mask"]), t(g["img_feat"])
    T, R = ids.shape[1], img.shape[1]
    seq, pooled, atts = O.bert_img_model(sd_g, "", cfg, ids, tt, im, img)
    close(seq, g["global_seq"], 5e-5); close(pooled, g["global_pooled"], 5e-5)
    close(atts[0], g["global_att0"]); close(atts[11], g["global_att11"], 5e-5)
    img_mask = torch.cat([im[:, :1], im[:, -R:]], dim=-1)
    seq, pooled, _ = O.bert_img_model(sd_g, "", cfg, ids[:, :1], None, img_mask, img)
    close(seq, g["imgonly_seq"], 5e-5); close(pooled, g["imgonly_pooled"], 5e-5)
    (seq, pooled, atts), ch = O.seq_bert_img_model(sd_s, "", cfg, ids, tt, t(g["chunk_attention_mask"]),
                                                   im, img, gi_list(g["gather_index"]))
    close(seq, g["seq_seq"], 5e-5); close(pooled, g["seq_pooled"], 5e-5)
    close(ch, g["chunk_hidden"], 5e-5)
    for i in (0, 5, 9, 11):
        close(atts[i], g["seq_att%d" % i], 5e-5)
    amap = torch.stack(atts[-3:], dim=1).sum(1).sum(1)[:, :T, T:]
    close(amap, g["align_map"], 5e-5)


def test_g13_seq_enc_residual_flags():
    """config.add_local_residual / config.add_residual (v10:212-223) against the reference run with the flags set"""
    g = H.load_golden("G13_seq_enc_residuals")
    rs = np.random.RandomState(int(g["seed"]))
    H.bert_img_weights(rs, _small_cfg())
    sd_s = H.to_torch(H.bert_img_weights(rs, _small_cfg(), seq=True))
    ids, tt, im, img = t(g["input_ids"]), t(g["token_type_ids"]), t(g["input_mask"]), t(g["img_feat"])
    for tag, local, resid in (("both", True, True), ("local", True, False), ("final", False, True)):
        cfg = dict(_small_cfg(), add_local_residual=local, add_residual=resid)
        (seq, pooled, atts), ch = O.seq_bert_img_model(sd_s, "", cfg, ids, tt, t(g["chunk_attention_mask"]), im, img,
                                                       gi_list(g["gather_index"]))
        close(seq, g[tag + "_seq"], 5e-5); close(pooled, g[tag + "_pooled"], 5e-5)
        close(ch, g[tag + "_chunk_hidden"], 5e-5); close(atts[11], g[tag + "_att11"], 5e-5)


def test_g6_calec_forward_and_head_grads():
    g = H.load_golden("G6_calec_small")
    cfg = _small_cfg()
    rs = np.random.RandomState(int(g["seed"]))
    sd = H.to_torch(H.calec_weights(rs, cfg, ""))
    for k, v in sd.items():
        if "enc." not in k:
            v.requires_grad_(True)
    cls, loss, _ = O.chunkalign_ensemble(
        sd, "", cfg, t(g["input_ids"]), t(g["img_feat"]), t(g["input_mask"]), t(g["token_type_ids"]),
        t(g["chunk_attention_mask"]), gi_list(g["gather_index"]), t(g["align_pos"]), t(g["total_label"]))
    close(cls, g["cls"], 5e-5)
    close(loss, g["align_loss"], 5e-5)
    (cls * t(g["dcls"])).sum().backward()
    have = sorted(k for k, v in sd.items() if v.grad is not None)
    # exactly the params the reference trains
    assert have == sorted(k[5:] for k in g["grad_names"].tolist())
    for k in g:
        if k.startswith("grad."):
            close(sd[k[5:]].grad, g[k], 1e-4)


def test_g10_enc4_align_losses_and_grads_through_both_encoders():
    """ChunkAlign_CLS_enc4_align (v10:1016-1084): classification loss, align loss and the gradient of their
    sum wrt head AND encoder parameters (the align loss reaches seq_enc through the attention probabilities)."""
    g = H.load_golden("G10_enc4_align")
    cfg = _small_cfg()
    rs = np.random.RandomState(int(g["seed"]))
    sd = H.to_torch(H.enc4_align_weights(rs, cfg, ""))
    for v in sd.values():
        if v.dtype.is_floating_point:
            v.requires_grad_(True)
    loss, matched, al = O.chunkalign_enc4_align(
        sd, "", cfg, t(g["input_ids"]), t(g["img_feat"]), t(g["input_mask"]), t(g["token_type_ids"]),
        t(g["chunk_attention_mask"]), gi_list(g["gather_index"]), t(g["label"]), t(g["align_pos"]), t(g["total_label"]))
    close(loss, g["loss_cls"], 2e-5)
    close(al, g["align_loss"], 2e-5)
    assert matched.to(torch.int64).tolist() == g["matched"].tolist()
    (loss + al).backward()
    have = sorted(k for k, v in sd.items() if v.grad is not None and float(v.grad.abs().max()) > 0)
    ref_names = sorted(g["grad_names"].tolist())
    assert set(have) <= set(ref_names)
    for k in g:
        if k.startswith("grad."):
            close(sd[k[5:]].grad, g[k], 2e-4)


def test_g7_cls_layer_lyx_forward_backward():
    g = H.load_golden("G7_cls_layer_lyx")
    rs = np.random.RandomState(int(g["seed"]))
    sd = {}
    H.cls_layer_lyx_weights(rs, sd, "", 128, 512)
    sd = {k: v.requires_grad_(True) for k, v in H.to_torch(sd).items()}
    kv, c = t(g["kv"]).requires_grad_(True), t(g["cls"]).requires_grad_(True)
    y = O.cls_layer_lyx(kv, c, sd, "", 1e-12)
    close(y, g["y"])
    (y * t(g["dy"])).sum().backward()
    close(kv.grad, g["dkv"], 2e-5); close(c.grad, g["dcls"], 2e-5)
    for k in g:
        if k.startswith("grad."):
            close(sd[k[5:]].grad, g[k], 2e-5)


def test_g8_abstract_specific_loss_logits_grads():
    g = H.load_golden("G8_abstract_specific")
    cfg = H.cfg_dict(hidden=768, heads=12, layers=12, vocab=2000, max_pos=64, img_dim=70)
    rs = np.random.RandomState(int(g["seed"]))
    sd = H.to_torch(H.abstract_specific_weights(rs, cfg))
    for k, v in sd.items():
        if "_enc." not in k:
            v.requires_grad_(True)
    batch = {k: t(g[k]) for k in ("input_ids", "token_type_ids", "input_mask", "img_feat",
                                  "chunk_attention_mask", "total_label", "align_pos", "label",
                                  "roberta_input_ids")}
    batch["gather_index"] = gi_list(g["gather_index"])
    loss, aux, logits, _ = O.abstract_specific(
        sd, cfg, batch, lambda ids, tt, m, pe, pm: H.stub_roberta_pooled(pe, ids))
    close(logits, g["logits"], 5e-5)
    close(loss, g["loss"], 5e-5)
    assert aux[0] is None and aux[1] is None and aux[3] is None
    loss.backward()
    have = sorted(k for k, v in sd.items() if v.grad is not None)
    assert have == sorted(g["grad_names"].tolist())
    for k in g:
        if k.startswith("grad."):
            close(sd[k[5:]].grad, g[k], 1e-4)
        if k.startswith("gsum."):
            got = sd[k[5:]].grad.abs().sum().item()
            assert abs(got - g[k][1]) <= 2e-3 * max(1e-3, g[k][1]), k


def test_mc_ce_matches_torch_probability_target_ce():
    logits = torch.randn(5, 4, dtype=torch.float64)
    label = torch.eye(4, dtype=torch.float64)[torch.tensor([0, 3, 1, 2, 2])]
    ref = torch.nn.CrossEntropyLoss()(logits, label)
    assert abs(O.mc_ce(logits, label).item() - ref.item()) < 1e-12


def test_g11_roberta_restatement_vs_stock_transformers_roberta():
    """oracle.roberta_prefix without prefix vectors against the STOCK transformers.RobertaModel (golden G11, tiny random
    config): position ids from the cumulative non-pad count, embeddings LayerNorm, eps 1e-5 layers, pooler.  The prefix
    splice itself stays this build's documented choice (the reference's module is absent)."""
    g = H.load_golden("G11_stock_roberta")
    n, t, h, a, layers = [int(v) for v in g["shape"]]
    sd = {k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w.")}
    cfg = dict(num_hidden_layers=layers, num_attention_heads=a, layer_norm_eps=1e-5, pad_token_id=1)
    seq, pooled = O.roberta_prefix(sd, "", cfg, torch.from_numpy(g["input_ids"]), torch.from_numpy(g["token_type_ids"]),
                                   torch.from_numpy(g["attention_mask"]), None, None)
    valid = torch.from_numpy(g["attention_mask"])[..., None]
    assert float(((seq - torch.from_numpy(g["seq"])) * valid).abs().max()) < 2e-5
    assert float((pooled - torch.from_numpy(g["pooled"])).abs().max()) < 2e-5
