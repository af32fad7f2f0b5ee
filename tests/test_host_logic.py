"""CPU tests of the host-side logic around the C ABI (no GPU, no HIP compute): batch contract,
chunk-id packing, trainable-parameter selection, state-dict key compatibility, the flat-gradient
all-reduce on a world_size-2 gloo group, and that the product path refuses to run without a GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import helpers as H

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "multimodal-context-reasoning_amd")


def test_synthetic_batch_honours_the_reference_contract():
    from Data import synthetic
    b = synthetic.make_batch(3, T=20, R=12, seed=5, vocab_size=30567, img_dim=70, min_text=6, min_regions=3,
                             roberta_len=16)
    n = 12
    assert b["input_ids"].shape == (n, 20) and b["input_ids"].dtype == torch.int64
    assert b["input_mask"].shape == (n, 32) and b["input_mask"].dtype == torch.float32
    assert b["img_feat"].shape == (n, 12, 70) and b["chunk_attention_mask"].shape == (n, 20, 20)
    assert b["label"].view(3, 4).sum(1).tolist() == [1.0, 1.0, 1.0]            # one-hot per example
    assert set(b) >= {"r_input_ids", "r_token_type_ids", "r_attention_mask", "offsets", "gather_index",
                      "total_label", "align_pos", "image", "text"}
    for i in range(n):
        ln = int(b["input_mask"][i, :20].sum())
        gi, offs = b["gather_index"][i], b["offsets"][i]
        assert gi.numel() == ln - 2                                            # tokens strictly inside CLS..SEP
        assert sorted(t for ch in offs for t in ch) == list(range(1, ln - 1))  # chunks partition 1..len-2
        cm = b["chunk_attention_mask"][i]
        assert cm[0, :ln].all() and cm[ln - 1, :ln].all() and cm[ln:, :].sum() == 0
        for k, ch in enumerate(offs):
            assert (gi[[t - 1 for t in ch]] == k).all()
            assert cm[ch[0], ch].all()
        assert ((b["total_label"][i] != 0).long() == b["align_pos"][i]).all()


def test_pack_chunk_ids_rows_and_padding():
    from modeling.modeling_vcr_chunkalign_v10 import pack_chunk_ids
    gi = [torch.tensor([0, 0, 1, 2, 2, 2]), torch.tensor([0, 1]), torch.zeros(0, dtype=torch.int64)]
    cid = pack_chunk_ids(gi, 10, torch.device("cpu"))
    assert cid.dtype == torch.int32 and cid.shape == (3, 10)
    assert cid[0].tolist() == [-1, 0, 0, 1, 2, 2, 2, -1, -1, -1]
    assert cid[1].tolist() == [-1, 0, 1] + [-1] * 7
    assert cid[2].tolist() == [-1] * 10


def test_host_packed_gather_index_equals_device_packing():
    from Data import synthetic
    from modeling import train_utils as tu
    from modeling.modeling_vcr_chunkalign_v10 import pack_chunk_ids
    b = synthetic.make_batch(2, T=16, R=6, seed=3, img_dim=70, min_text=6, min_regions=3, roberta_len=8)
    host = tu.pack_gather_index(b["gather_index"], 16)
    assert torch.equal(host, pack_chunk_ids(b["gather_index"], 16, torch.device("cpu")))
    moved = tu.batch_to_device(b, torch.device("cpu"))
    assert torch.is_tensor(moved["gather_index"]) and moved["gather_index"].dtype == torch.int32


def _tiny_model():
    from modeling.bert_primitives import BertConfig
    from modeling.modeling_ensemble import Abstract_Specific
    from modeling.modeling_transfomres import BertImgModel
    from modeling.modeling_vcr_chunkalign_v10 import ChunkAlign_CLS_enc4_align_ensemble, SeqBertImgModel
    from modeling.roberta_prefix import PrefixPoolerStandIn
    cfg = BertConfig(vocab_size=200, num_hidden_layers=2, max_position_embeddings=32, img_feature_dim=70)
    calec = ChunkAlign_CLS_enc4_align_ensemble(BertImgModel(cfg), SeqBertImgModel(cfg), 4)
    return Abstract_Specific(calec_model=calec, clip_model=None, roberta_model=PrefixPoolerStandIn(), num_labels=4)


def test_state_dict_keys_match_the_reference_and_trainable_set():
    """Key names were pinned against the reference by tools/gen_golden.py (strict load); here the
    drop-in model must expose exactly that key set (+ the stand-in's roberta.*)."""
    from modeling import train_utils as tu
    model = _tiny_model()
    cfg = H.cfg_dict(hidden=768, heads=12, layers=2, vocab=200, max_pos=32, img_dim=70)
    ref_keys = set(H.abstract_specific_weights(np.random.RandomState(0), cfg))
    mine = {k for k in model.state_dict() if not k.endswith("position_ids") and not k.startswith("roberta.")}
    assert mine == ref_keys, sorted(mine ^ ref_keys)[:10]
    names = tu.trainable_parameters(model)
    g8 = H.load_golden("G8_abstract_specific")
    ref_trainable = set(g8["grad_names"].tolist())                 # what the reference's backward reaches
    assert {n for n in names if not n.startswith("roberta.")} == ref_trainable


def test_error_conventions_of_the_reference_classes():
    """SURVEY 8b, return / error conventions -- raised before any device work, so checkable without a GPU: NotImplementedError for
    attention masks of the wrong rank (modeling_transfomres.py:628-641, v10:289-312) and for a head_mask, the `assert img_feats is
    None` with history states (modeling_transfomres.py:661-662), ValueError for an embed_dim the head count does not divide
    (v10:704-708), NotImplementedError for the call patterns of cross_attention_lyx ModCR never uses."""
    from modeling.modeling_vcr_chunkalign_v10 import cross_attention_lyx
    model = _tiny_model()
    gm, sm = model.calec.global_enc, model.calec.seq_enc
    ids = torch.ones(2, 6, dtype=torch.int64)
    img = torch.zeros(2, 3, 70)
    # (a 3-D mask on global_enc is accepted since round 4, as modeling_transfomres.py:629-630 does: tests/test_hip_models.py::
    # test_bert_img_model_takes_a_3d_attention_mask; ranks 1 and 4 still raise)
    with pytest.raises(NotImplementedError):
        gm(ids, img_feats=img, attention_mask=torch.ones(9))                             # rank 1
    with pytest.raises(NotImplementedError):
        gm(ids, img_feats=img, attention_mask=torch.ones(2, 1, 1, 9))                    # rank 4
    with pytest.raises(NotImplementedError):
        gm(ids, img_feats=img, attention_mask=torch.ones(2, 9), head_mask=torch.ones(2, 12))
    with pytest.raises(AssertionError):
        gm(ids, img_feats=img, attention_mask=torch.ones(2, 9), encoder_history_states=[torch.zeros(2, 2, 768)] * 2)
    with pytest.raises(NotImplementedError):
        sm(ids, img_feats=img, attention_mask=torch.ones(2, 9), input_mask=torch.ones(2, 9))            # seq_enc wants the 3-D chunk mask
    with pytest.raises(NotImplementedError):
        sm(ids, img_feats=img, attention_mask=torch.ones(2, 6, 6), input_mask=None)
    with pytest.raises(NotImplementedError):
        sm(ids, img_feats=img, attention_mask=torch.ones(2, 6, 6), input_mask=torch.ones(2, 9), head_mask=torch.ones(2, 12))
    with pytest.raises(ValueError):
        cross_attention_lyx(100, 8)
    xa = cross_attention_lyx(64, 8)
    with pytest.raises(NotImplementedError):
        xa(torch.zeros(2, 1, 64), key_value_states=None)
    with pytest.raises(NotImplementedError):
        xa(torch.zeros(2, 1, 64), key_value_states=torch.zeros(2, 5, 64), attention_mask=torch.zeros(2, 1, 1, 5))
    with pytest.raises(NotImplementedError):
        xa(torch.zeros(2, 3, 64), key_value_states=torch.zeros(2, 5, 64))


def test_vcr_script_freezes_the_roberta_body_and_uses_vcr_defaults():
    """run_vcr_ModCR.py:781-787: every RoBERTa parameter whose name contains neither 'embeddings.' nor 'pooler.' is frozen, so
    the step's trainable set is the heads + roberta.embeddings.* + roberta.pooler.*; :487-533: vcr_data/ file defaults."""
    import run_PMR_ModCR as pmr
    import run_vcr_ModCR as vcr
    from modeling import train_utils as tu
    from modeling.roberta_prefix import RobertaPrefixModel
    model = _tiny_model()
    model.roberta = RobertaPrefixModel(vocab_size=300, hidden_size=1024, num_hidden_layers=2, num_attention_heads=16, intermediate_size=256,
                                       max_position_embeddings=40)
    before = set(tu.trainable_parameters(model))
    assert "roberta.encoder.layer.1.attention.self.query.weight" in before
    frozen = vcr.freeze_roberta_body(model)
    assert frozen and all(n.startswith("encoder.layer.") for n in frozen)
    names = set(tu.trainable_parameters(model))
    rob = {n for n in names if n.startswith("roberta.")}
    assert rob == {"roberta." + n for n, _ in model.roberta.named_parameters() if "embeddings." in n or "pooler." in n}
    assert rob and {n for n in names if not n.startswith("roberta.")} == {n for n in before if not n.startswith("roberta.")}
    # the reference's own rule, restated on the reference's names
    ref_rule = {n for n, _ in model.roberta.named_parameters() if not ('embeddings.' not in n and 'pooler.' not in n)}
    assert {n[len("roberta."):] for n in rob} == ref_rule
    # defaults: the wrapper hands run_PMR_ModCR the VCR file names, batch 8 x 4, validation every 3500 steps
    seen = {}
    real_main = pmr.main
    pmr.main = lambda argv: seen.setdefault("args", pmr.get_args(argv))
    try:
        vcr.main(["--learning_rate", "2e-5"])
    finally:
        pmr.main = real_main
    a = seen["args"]
    assert a.roberta_file_train == "vcr_data/vcr_train_CALeC.pkl" and a.vcr_chunk_mask_test == "vcr_data/ChunkMaskTest_v4_vcr.pkl"
    assert a.vcr_feat_file_dev == "vcr_data/image_feature/val_feat_vcr.pkl" and a.clip_file_dev == "vcr_data/vcr_val.json"
    assert (a.per_gpu_train_batch_size, a.gradient_accumulation_steps, a.valid_steps) == (8, 4, 3500) and a.learning_rate == 2e-5
    assert pmr.MODEL_HOOKS == [] and pmr.CKPT_TAG == "VCR-Prefix-tuning_len5_all"
    pmr.CKPT_TAG = "Multi-View-Reasoning-Prefix-tuning_LV_3_LA_7"


def test_enc4_align_keys_and_trainable_encoder_set():
    """ChunkAlign_CLS_enc4_align (v10:1016-1027): state-dict keys equal the reference's (the G10 golden was produced through a
    strict load of exactly these keys), and set_train_encoders() on the ensemble registers every encoder parameter the
    forward uses (not the unused edge_dense) with the gradient buffers / optimizer."""
    from modeling import train_utils as tu
    from modeling.bert_primitives import BertConfig
    from modeling.modeling_transfomres import BertImgModel
    from modeling.modeling_vcr_chunkalign_v10 import SeqBertImgModel, ChunkAlign_CLS_enc4_align
    cfgd = H.cfg_dict(hidden=128, heads=2, layers=2, vocab=200, max_pos=32, img_dim=70)
    cfg = BertConfig(hidden_size=128, num_attention_heads=2, intermediate_size=512, num_hidden_layers=2, vocab_size=200,
                     max_position_embeddings=32, img_feature_dim=70, img_feature_type="frcnn", use_img_layernorm=1,
                     img_layer_norm_eps=1e-12, output_attentions=True, max_hypo=50, add_residual=False, add_local_residual=False)
    m = ChunkAlign_CLS_enc4_align(BertImgModel(cfg), SeqBertImgModel(cfg), 4)
    ref_keys = set(H.enc4_align_weights(np.random.RandomState(0), cfgd, ""))
    mine = {k for k in m.state_dict() if not k.endswith("position_ids")}
    assert mine == ref_keys, sorted(mine ^ ref_keys)[:10]
    assert m.global_enc.trainable and m.seq_enc.trainable
    model = _tiny_model()
    base = set(tu.trainable_parameters(model))
    model.calec.set_train_encoders(True)
    names = set(tu.trainable_parameters(model))
    enc = {k for k in names - base}
    assert enc and all(k.startswith("calec.global_enc.") or k.startswith("calec.seq_enc.") for k in enc)
    assert "calec.seq_enc.edge_dense.weight" not in names
    assert "calec.seq_enc.encoder.layer.1.attention.self.query.weight" in names and "calec.global_enc.img_embedding.weight" in names


def test_product_path_refuses_to_run_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    r = subprocess.run([sys.executable, os.path.join(PKG, "run_PMR_ModCR.py"), "--do_eval"], capture_output=True,
                       text=True, cwd=PKG, timeout=600)
    assert r.returncode != 0 and "no CPU fallback" in (r.stderr + r.stdout)


def test_flat_gradients_lay_qkv_of_a_block_back_to_back_when_named():
    """FlatGrads(names=...): the query | key | value weights (and biases) of one attention block are one span of the buffer (the
    [3H, H] weight-gradient product of the attention backward writes it in place, hip_autograd.BertLayerFn.backward); everything else
    keeps the reverse registration order; take_span hands a span out once per accumulation window."""
    sys.path.insert(0, PKG)
    from modeling import train_utils as tu
    h = 64
    names, params = [], []
    for layer in range(2):
        pre = "enc.layer.%d." % layer
        for q in ("query", "key", "value"):
            names += [pre + "attention.self.%s.weight" % q, pre + "attention.self.%s.bias" % q]
            params += [torch.nn.Parameter(torch.randn(h, h)), torch.nn.Parameter(torch.randn(h))]
        names += [pre + "attention.output.dense.weight", pre + "attention.output.dense.bias"]
        params += [torch.nn.Parameter(torch.randn(h, h)), torch.nn.Parameter(torch.randn(h))]
    names.append("head.bias")
    params.append(torch.nn.Parameter(torch.randn(1)))
    flat = tu.FlatGrads(params, torch.device("cpu"), names=names)
    pd = dict(zip(names, params))
    off = lambda k: flat.offsets[id(pd[k])]
    for layer in range(2):
        pre = "enc.layer.%d.attention.self." % layer
        assert off(pre + "key.weight") == off(pre + "query.weight") + h * h and off(pre + "value.weight") == off(pre + "query.weight") + 2 * h * h
        assert off(pre + "key.bias") == off(pre + "query.bias") + h and off(pre + "value.bias") == off(pre + "query.bias") + 2 * h
        assert off(pre + "query.bias") == off(pre + "value.weight") + h * h
    assert off("head.bias") == 0 and off("enc.layer.1.attention.output.dense.bias") == 64            # reverse order, 64-element starts
    assert off("enc.layer.1.attention.self.query.weight") < off("enc.layer.0.attention.output.dense.bias")
    assert [flat.offsets[id(p)] for p in flat.order] == sorted(flat.offsets.values())
    assert all(p.grad.data_ptr() == flat.flat.data_ptr() + 4 * flat.offsets[id(p)] for p in params)
    trio = [pd["enc.layer.0.attention.self.%s.weight" % q] for q in ("query", "key", "value")]
    span = flat.take_span(trio)
    assert span is not None and span.numel() == 3 * h * h and span.data_ptr() == trio[0].grad.data_ptr()
    assert flat.take_span(trio) is None                      # written once in this window: a second micro-batch goes through autograd
    flat.zero()
    assert flat.take_span(trio) is not None
    assert flat.take_span([trio[0], trio[2]]) is None        # not adjacent
    # without names: plain reverse order, no spans
    flat2 = tu.FlatGrads(params, torch.device("cpu"))
    assert flat2.take_span(trio) is None


def _ddp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, PKG)
    from modeling import train_utils as tu
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(5, 3)), torch.nn.Parameter(torch.randn(7))]
    flat = tu.FlatGrads(params, torch.device("cpu"))
    # rank-dependent "local" gradients accumulated the way autograd does (in place on the views)
    for i, p in enumerate(params):
        p.grad.add_(torch.full_like(p, float(rank + 1 + i)))
    flat.all_reduce(world)
    opt = torch.optim.SGD(params, lr=1.0)
    before = [p.detach().clone() for p in params]
    opt.step()
    ok = all(torch.allclose(p.grad, torch.full_like(p, (1 + 2) / 2 + i)) for i, p in enumerate(params))
    ok = ok and all(torch.allclose(b - p.detach(), p.grad) for b, p in zip(before, params))
    ok = ok and params[-1].grad.data_ptr() == flat.flat.data_ptr()           # still views of the flat buffer (reverse order)
    flat.zero()
    ok = ok and all(float(p.grad.abs().sum()) == 0 for p in params)
    q.put((rank, ok, [p.detach().sum().item() for p in params]))
    dist.barrier()
    dist.destroy_process_group()


def test_flat_gradient_all_reduce_world_size_2_gloo():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_ddp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == res[1][2]              # identical parameters on both ranks after the step


def _ddp_overlap_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, PKG)
    from modeling import train_utils as tu
    torch.manual_seed(0)                                   # same weights on both ranks
    net = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.Tanh(), torch.nn.Linear(32, 32), torch.nn.Tanh(),
                              torch.nn.Linear(32, 3), torch.nn.Linear(3, 3))
    unused = torch.nn.Parameter(torch.zeros(4))            # registered but never reached by backward
    params = list(net.parameters()) + [unused]
    flat = tu.FlatGrads(params, torch.device("cpu"), bucket_bytes=512)
    nb = len(flat.buckets)
    g = torch.Generator().manual_seed(100 + rank)          # different data per rank
    x = torch.randn(16, 6, generator=g)
    # reference: plain single all-reduce of the same local gradients
    ref = []
    net(x).square().mean().backward()
    flat.all_reduce(world)
    ref = flat.flat.clone()
    flat.zero()
    flat.begin(world)
    net(x).square().mean().backward()
    launched = flat.launched_in_backward
    flat.finish(world)
    ok = torch.allclose(flat.flat, ref, rtol=0, atol=1e-7) and float(ref.abs().sum()) > 0
    ok = ok and nb >= 3 and 1 <= launched < nb             # the bucket holding `unused` is launched by finish()
    ok = ok and all(p.grad.data_ptr() >= flat.flat.data_ptr() for p in params)
    # opt-in bf16 buckets: the same hooks, every bucket reduced as a bf16 copy -- the mean gradient to bf16's
    # 8 bits of every summand, still bit-identical on the two ranks (checked through the sum the parent compares)
    flat16 = tu.FlatGrads(params, torch.device("cpu"), bucket_bytes=512, comm_dtype=torch.bfloat16)
    flat16.begin(world)
    net(x).square().mean().backward()
    flat16.finish(world)
    err = float((flat16.flat - ref).abs().max()) / float(ref.abs().max())
    ok = ok and 0 < err <= 2 ** -7 and flat16.flat.dtype == torch.float32 and not flat16._comm
    q.put((rank, ok, float(flat.flat.sum()) + float(flat16.flat.double().sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_all_reduce_overlapped_with_backward_world_size_2_gloo():
    """The gradient buckets launched from the post-accumulate hooks while backward runs give exactly the single
    all-reduce's result; a bucket whose parameter got no gradient is reduced by finish()."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30100 + (os.getpid() % 500)
    procs = [ctx.Process(target=_ddp_overlap_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == res[1][2]


def _ddp_accum_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, PKG)
    from modeling import train_utils as tu
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.Tanh(), torch.nn.Linear(32, 3))
    params = list(net.parameters())
    flat = tu.FlatGrads(params, torch.device("cpu"), bucket_bytes=512)
    g = torch.Generator().manual_seed(200 + rank)
    xs = [torch.randn(16, 6, generator=g) * 3 for _ in range(3)]
    max_norm = 0.05
    # what DistributedDataParallel + clip_grad_norm_ after every backward computes (reference run_PMR_ModCR.py:203-216):
    # g <- clip(g + mean over ranks of the micro-batch's gradient), micro-step by micro-step
    want = torch.zeros_like(flat.flat)
    for x in xs:
        flat.zero()
        net(x).square().mean().backward()
        local = flat.flat.clone()
        dist.all_reduce(local)
        want = want + local / world
        want = want * min(1.0, max_norm / (float(want.norm()) + 1e-6))
    # the run scripts' loop: begin / backward / finish at EVERY micro-step on the accumulating buffer, then the clip
    flat.zero()
    for x in xs:
        flat.begin(world)
        net(x).square().mean().backward()
        flat.finish(world)
        torch.nn.utils.clip_grad_norm_(flat.params, max_norm)
    ok = torch.allclose(flat.flat, want, rtol=1e-5, atol=1e-7) and float(want.abs().sum()) > 0
    q.put((rank, ok, float(flat.flat.sum())))
    dist.barrier()
    dist.destroy_process_group()


def test_accumulated_micro_steps_are_reduced_before_each_clip_world_size_2_gloo():
    """gradient_accumulation_steps > 1 with world_size > 1 (the VCR configuration: 4 x 8): every micro-step's gradient is
    all-reduced before the per-micro-step clip, as under the reference's DistributedDataParallel."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 30700 + (os.getpid() % 500)
    procs = [ctx.Process(target=_ddp_accum_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(60)
    assert all(ok for _, ok, _ in res)
    assert res[0][2] == res[1][2]


@pytest.mark.parametrize("seed", [3, 17])
def test_host_collate_matches_the_reference_collate(seed):
    """Data/collate.py::SNLIGPT_gen_collate against the REFERENCE's own SNLIGPT_gen_collate
    (Data/VCRChunkAlign.py:690-741) run on the same ragged samples (golden G12, tools/gen_golden.py::g12_reference_collate):
    every tensor of the batch dict bit for bit, dtypes, key set, the ragged lists; plus the host-packed chunk-id rows."""
    from Data import collate as C
    g = H.load_golden("G12_reference_collate_seed%d" % seed)
    examples = H.collate_samples(seed)
    b = C.SNLIGPT_gen_collate(examples)
    assert set(g["keys"].tolist()) == set(b.keys()) - {"gather_index_list"}
    for k in ("r_input_ids", "r_token_type_ids", "r_attention_mask", "input_ids", "token_type_ids", "input_mask", "img_feat", "label",
              "chunk_attention_mask", "total_label", "align_pos"):
        ref = torch.from_numpy(g[k])
        assert b[k].shape == ref.shape, (k, b[k].shape, ref.shape)
        assert torch.equal(b[k].to(ref.dtype), ref), k
    assert str(b["label"].dtype) == str(g["label_dtype"]) and str(b["input_mask"].dtype) == str(g["input_mask_dtype"])
    assert b["input_ids"].dtype == torch.int64 and b["total_label"].dtype == torch.int64
    assert b["image"] is None and b["text"] is None and bool(g["image_is_none"])
    assert list(g["img_id"]) == b["img_id"] and list(g["ques_str"]) == b["ques_str"] and list(g["ans_str"]) == b["ans_str"]
    assert [len(o) for o in b["offsets"]] == g["n_offsets"].tolist()
    t = b["input_ids"].shape[1]
    for i, row in enumerate(g["gather_index"]):
        gi = torch.from_numpy(row[row >= 0])
        assert torch.equal(b["gather_index_list"][i], gi)
        packed = b["gather_index"][i]
        assert packed.dtype == torch.int32 and int(packed[0]) == -1
        assert torch.equal(packed[1:1 + gi.numel()].to(torch.int64), gi) and bool((packed[1 + gi.numel():] == -1).all())
    from modeling import train_utils as tu
    assert torch.equal(tu.pack_gather_index(b["gather_index_list"], t), b["gather_index"])
    # pack=False returns the reference's own type for gather_index (list of ragged int64 tensors)
    b2 = C.SNLIGPT_gen_collate(examples, pack=False)
    assert isinstance(b2["gather_index"], list) and set(b2.keys()) == set(g["keys"].tolist())


def test_data_parallel_shards_are_disjoint_and_cover():
    """DistributedSampler-style sharding of whole examples (the 4 choices stay on one rank)."""
    from torch.utils.data.distributed import DistributedSampler
    from Data.synthetic import SyntheticPMRDataset
    ds = SyntheticPMRDataset(64, T=12, R=6, img_dim=70)
    seen = []
    for rank in range(2):
        s = DistributedSampler(ds, num_replicas=2, rank=rank, shuffle=True, seed=0)
        s.set_epoch(0)
        seen.append(set(iter(s)))
    assert not (seen[0] & seen[1]) and (seen[0] | seen[1]) == set(range(64))


def test_hf_adamw_restatement_properties_and_host_class():
    """oracle.hf_adamw_step restates transformers 4.x AdamW.step (absent from transformers 5.15 and from the reference
    tree).  Pins available without the class: (a) a hand-computed first step, (b) with eps -> 0 it is torch.optim.Adam
    (same published algorithm up to where eps sits), (c) the non-fused product class HFAdamW equals it bit for bit in
    fp64, (d) with eps = 1e-5 and gradients ~1e-4 it is NOT torch.optim.AdamW."""
    from oracle import modcr_oracle as O
    from modeling import train_utils as tu
    torch.manual_seed(0)
    # (a) t = 1: exp_avg = 0.1 g, exp_avg_sq = 0.001 g^2, step = lr sqrt(0.001) / 0.1, denom = sqrt(0.001)|g| + eps
    p, g, st = torch.tensor([1.0, -2.0]), torch.tensor([0.5, -1e-4]), {}
    out = O.hf_adamw_step(p.clone(), g, st, lr=0.1, eps=1e-5)
    step = 0.1 * (0.001 ** 0.5) / 0.1
    hand = p - step * (0.1 * g) / ((0.001 * g * g).sqrt() + 1e-5)
    assert torch.allclose(out, hand, atol=1e-7) and st["step"] == 1
    # (b) eps -> 0: torch Adam
    w0 = torch.randn(17, dtype=torch.float64)
    grads = [torch.randn(17, dtype=torch.float64) for _ in range(5)]
    a = w0.clone(); sa = {}
    tp = torch.nn.Parameter(w0.clone())
    topt = torch.optim.Adam([tp], lr=1e-2, eps=1e-30)
    for gq in grads:
        O.hf_adamw_step(a, gq, sa, lr=1e-2, eps=1e-30)
        tp.grad = gq.clone(); topt.step()
    assert torch.allclose(a, tp.detach(), rtol=1e-10, atol=1e-12)
    # (c) HFAdamW class == restatement; (d) != torch.optim.AdamW in the small-gradient regime
    q = torch.nn.Parameter(w0.clone()); r = torch.nn.Parameter(w0.clone())
    hopt = tu.HFAdamW([q], lr=1e-3, eps=1e-5); ropt = torch.optim.AdamW([r], lr=1e-3, eps=1e-5, weight_decay=0.0)
    b = w0.clone(); sb = {}
    for gq in grads:
        gq = gq * 1e-4
        O.hf_adamw_step(b, gq, sb, lr=1e-3, eps=1e-5)
        q.grad = gq.clone(); hopt.step()
        r.grad = gq.clone(); ropt.step()
    assert torch.equal(q.detach(), b)
    assert float((r.detach() - b).abs().max()) > 50 * 1e-3 * 1e-2      # first steps come out several times larger in torch's form


def test_hf_adamw_reference_layout_round_trip():
    """HFAdamW (the non-fused route) writes / reads the reference's optimizer layout -- indices in the reference's parameter
    order over ALL parameters, two groups -- so its resume files are interchangeable with FlatAdamW's."""
    from modeling import train_utils as tu
    model = _tiny_model()
    names = tu.trainable_parameters(model)
    opt, _ = tu.make_optimizer(model, names, learning_rate=1e-3, t_total=10)
    torch.manual_seed(0)
    pd = dict(model.named_parameters())
    for n in names:
        pd[n].grad = torch.randn_like(pd[n]) * 0.01
    opt.step()
    sd = opt.reference_state_dict(model)
    g0, g1 = tu.reference_param_order(model)
    assert [len(g["params"]) for g in sd["param_groups"]] == [len(g0), len(g1)]
    assert sorted((g0 + g1)[i] for i in sd["state"]) == sorted(names)
    assert abs(sd["param_groups"][1]["lr"] - 0.1 * sd["param_groups"][0]["lr"]) < 1e-12
    opt2, _ = tu.make_optimizer(model, names, learning_rate=1e-3, t_total=10)
    opt2.load_reference_state_dict(sd, model)
    for n in names:
        a, b = opt.state[pd[n]], opt2.state[pd[n]]
        assert a["step"] == b["step"] == 1 and torch.equal(a["exp_avg"], b["exp_avg"]) and torch.equal(a["exp_avg_sq"], b["exp_avg_sq"])


def test_schedules_equal_the_transformers_schedules():
    """run_PMR_ModCR.py:138-145 picks transformers.get_linear_schedule_with_warmup / get_constant_schedule_with_warmup:
    both still exist in the installed transformers, so train_utils.lr_lambda and the oracle's restatement are pinned
    against the real functions."""
    import transformers
    from oracle import modcr_oracle as O
    from modeling import train_utils as tu
    for warm, total in ((0, 10), (3, 10), (0, 1), (5, 4)):
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([p], lr=1.0)
        sch = transformers.get_linear_schedule_with_warmup(opt, num_warmup_steps=warm, num_training_steps=total)
        lam = tu.lr_lambda("linear", warm, total)
        for step in range(total + 3):
            assert abs(opt.param_groups[0]["lr"] - lam(step)) < 1e-12 and abs(lam(step) - O.linear_schedule(step, total, warm)) < 1e-12
            opt.step(); sch.step()
        opt = torch.optim.SGD([p], lr=1.0)
        sch = transformers.get_constant_schedule_with_warmup(opt, num_warmup_steps=warm)
        lam = tu.lr_lambda("constant", warm, total)
        for step in range(8):
            assert abs(opt.param_groups[0]["lr"] - lam(step)) < 1e-12 and abs(lam(step) - O.constant_schedule(step, warm)) < 1e-12
            opt.step(); sch.step()
    with pytest.raises(ValueError):
        tu.lr_lambda("cosine", 0, 10)


def test_bench_spawn_returns_promptly_when_one_rank_dies():
    """a rank that dies early (OOM, missing .so, RCCL init) must not leave `bench.py --gpus N` waiting for the
    others' rendezvous timeout: the parent polls, kills the survivors and returns the failing code."""
    import argparse
    import time
    sys.path.insert(0, ROOT)
    import bench
    child = "import os, sys, time\nif os.environ['RANK'] == '0': sys.exit(3)\ntime.sleep(600)\n"
    t0 = time.time()
    rc = bench.spawn_ranks(argparse.Namespace(gpus=3), cmd=[sys.executable, "-c", child])
    assert rc == 3 and time.time() - t0 < 30
    ok = "import os\nassert os.environ['WORLD_SIZE'] == '2' and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
    assert bench.spawn_ranks(argparse.Namespace(gpus=2), cmd=[sys.executable, "-c", ok]) == 0


def test_bench_starts_its_own_ranks_without_a_launcher():
    """`python bench.py --gpus 2` (the driver's command shape for N > 1 without torch.distributed.run) must start the two
    ranks itself instead of exiting on WORLD_SIZE.  No GPU here: each child reports the missing
    device and the parent returns their failure -- what is checked is that BOTH ranks were started with a rendezvous env."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK") and not k.startswith("MODCR_")}
    if torch.cuda.is_available():
        pytest.skip("GPU box: the bench itself covers this path")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode != 0
    # (the first rank to fail takes the other down at once, so the second line may be missing)
    assert "rank 0 of 2" in r.stderr or "rank 1 of 2" in r.stderr, r.stderr[-2000:]
    # and a knob in the environment is refused before anything else happens
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], capture_output=True, text=True,
                       env=dict(env, MODCR_FFN_SPLIT="2"), timeout=600)
    assert r.returncode != 0 and "refusing to run with tuning knobs" in r.stderr


def test_flat_grads_sink_counts_down_after_the_last_use():
    """FlatGrads.done(): a parameter with two forward nodes in the graph is counted down (its bucket launched, at N > 1) only when the
    second node has reported -- or when autograd's hook fires; finish() starts the next graph's count afresh; note_use ignores
    forwards under torch.no_grad() (hip_autograd._note_uses)."""
    from modeling import hip_autograd as ag
    from modeling import train_utils as tu
    p1, p2 = torch.nn.Parameter(torch.zeros(8)), torch.nn.Parameter(torch.zeros(8))
    flat = tu.FlatGrads([p1, p2], torch.device("cpu"), bucket_bytes=1)          # one bucket per parameter
    launched = []

    class Done(object):
        def wait(self):
            pass

    def rec(b):
        launched.append(b)
        flat._works[b] = Done()
    flat._launch = rec
    assert ag.grad_sink() is flat
    flat.zero()
    flat.note_use(p1); flat.note_use(p1); flat.note_use(p2)
    flat.begin(1, force=True)
    assert flat.take(p1, accumulates=True) is not None
    flat.done(p1)
    assert launched == []                                   # one of p1's two uses is still out
    assert flat.take(p2) is not None
    flat.done(p2)
    assert launched == [flat._bucket_of[id(p2)]]
    assert flat.take(p1, accumulates=True) is not None
    flat.done(p1)
    assert launched == [flat._bucket_of[id(p2)], flat._bucket_of[id(p1)]]
    flat.finish(1)
    # next graph: one use each; a written slice is not handed out again before zero(), an accumulating one is
    flat.note_use(p1); flat.note_use(p2)
    flat.begin(1, force=True)
    assert flat.take(p2) is None and flat.take(p1, accumulates=True) is not None
    flat.done(p1)
    assert launched[-1] == flat._bucket_of[id(p1)] and len(launched) == 3
    flat._on_grad(p2)                                       # autograd's hook for the parameter that went through autograd
    assert len(launched) == 4
    flat.finish(1)
    # no_grad forwards are not uses

    class Ctx(object):
        needs_input_grad = (False, True)
    lin = torch.nn.Linear(64, 64)
    flat2 = tu.FlatGrads(list(lin.parameters()), torch.device("cpu"))
    x = torch.randn(4, 64)
    import modcr_hip as mh
    if getattr(mh, "_lib", None) is None and not torch.cuda.is_available():
        # no GPU here: the forward of LinearFn would call the C ABI, so only the bookkeeping is exercised
        with torch.no_grad():
            ag._caller_grad_mode().append(torch.is_grad_enabled())
            ag._note_uses(Ctx(), ((1, lin.weight),))
            ag._caller_grad_mode().pop()
        assert flat2._uses.get(id(lin.weight), 0) == 0
        ag._caller_grad_mode().append(torch.is_grad_enabled())
        ag._note_uses(Ctx(), ((1, lin.weight),))
        ag._caller_grad_mode().pop()
        assert flat2._uses.get(id(lin.weight), 0) == 1
        # a second grad-enabled forward whose backward never runs would leave a count of 2 (single_use takes refused, done() never
        # reaching the use count): micro_step opens every training forward with new_graph(), which drops the dead graph's counts
        ag._caller_grad_mode().append(torch.is_grad_enabled())
        ag._note_uses(Ctx(), ((1, lin.weight),))
        ag._caller_grad_mode().pop()
        assert flat2._uses.get(id(lin.weight), 0) == 2 and flat2.take(lin.weight, single_use=True) is None
        flat2.new_graph()
        assert flat2._uses.get(id(lin.weight), 0) == 0
        ag._caller_grad_mode().append(torch.is_grad_enabled())
        ag._note_uses(Ctx(), ((1, lin.weight),))
        ag._caller_grad_mode().pop()
        assert flat2.take(lin.weight, single_use=True) is not None
    del x
    flat2.close()
    flat.install()
    # weak reference: a dropped buffer is no longer the sink
    flat.close()
    assert ag.grad_sink() is None
    flat.install()
    assert ag.grad_sink() is flat
    del flat
    import gc
    gc.collect()
    assert ag.grad_sink() is None


def test_backward_memory_auto_counts_every_group_of_trainable_layers(monkeypatch):
    """--modcr_backward_memory auto sizes what the trainable layers keep from the BUILT model -- 12 + 12 Oscar layers over
    text + regions when the encoders are trained, the RoBERTa body over its own length and width -- not from one layer."""
    from modeling import hip_layers
    from modeling import train_utils as tu
    model = tu.build_model(torch.device("cpu"), seed=0, hidden_size=128, num_hidden_layers=12, num_attention_heads=2, vocab_size=200,
                           max_position_embeddings=64, img_feature_dim=70, train_encoders=True)
    groups = hip_layers.backward_memory_groups(model, sequences=512, text_len=80, regions=100)
    assert groups == [(12, 512, 180, 128), (12, 512, 180, 128)]
    frozen = tu.build_model(torch.device("cpu"), seed=0, hidden_size=128, num_hidden_layers=12, num_attention_heads=2, vocab_size=200,
                            max_position_embeddings=64, img_feature_dim=70)
    assert hip_layers.backward_memory_groups(frozen, 512, 80, 100) == []
    need = sum(l * (n * 3 * 192 * h * 2 + n * s * 4 * h * 2) for l, n, s, h in groups)
    keep0 = (hip_layers.SAVE_QKV, hip_layers.KEEP_GELU_INPUT)
    try:
        monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (2 * need + 1024, 4 * need))
        assert hip_layers.configure_backward_memory("auto", device="cuda:0", groups=groups) == "keep"
        monkeypatch.setattr(torch.cuda, "mem_get_info", lambda dev=None: (2 * need - 1024, 4 * need))
        assert hip_layers.configure_backward_memory("auto", device="cuda:0", groups=groups) == "recompute"
        assert not hip_layers.SAVE_QKV and not hip_layers.KEEP_GELU_INPUT
        # one group of 12 layers alone would have fitted: the estimate really is the sum
        assert hip_layers.configure_backward_memory("auto", device="cuda:0", groups=groups[:1]) == "keep"
        assert hip_layers.configure_backward_memory("auto", device="cuda:0", groups=[]) == "keep"
    finally:
        hip_layers.SAVE_QKV, hip_layers.KEEP_GELU_INPUT = keep0
