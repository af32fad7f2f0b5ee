#!/usr/bin/env python
"""RCCL's first contact with FlatGrads, on ONE GPU: a process group over the `nccl` backend (= RCCL on ROCm) with world_size 1,
the bucketed all-reduces launched from the post-accumulate-grad hooks while the HIP backward is still running, and finish()
waiting for them -- the stream semantics and hook ordering of the N > 1 path (which the gloo tests cannot show), with a
collective that is an identity, so the reduced gradients must equal those of a step without any collective.
Prints one JSON line; exit code 0 = agreement."""
import json
import os
import socket
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402
from Data import synthetic  # noqa: E402
from modeling import train_utils as tu  # noqa: E402


def main():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
    train_enc = "--train-encoders" in sys.argv
    # --with-roberta: the reference's real step -- the trainable 24-layer prefix RoBERTa-large body: 1.66 GB of fp32 gradients in the
    # bucket plan bench.py / the run scripts use (64 MB buckets -> 26), launched from the hooks as its layers' backward finishes
    with_rob = "--with-roberta" in sys.argv
    model = tu.build_model(dev, seed=0, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1, train_encoders=train_enc,
                           **(dict(roberta_body="large", roberta_hidden_dropout_prob=0.1) if with_rob else {}))
    for t in list(model.parameters()) + list(model.buffers()):
        dist.broadcast(t.data, 0)                       # bench.py / run scripts: every rank starts from rank 0's weights
    model.train()
    names = tu.trainable_parameters(model)
    pd = dict(model.named_parameters())
    for k, p in pd.items():
        p.requires_grad_(k in names)
    flat = tu.FlatGrads([pd[k] for k in names], dev, bucket_bytes=(64 << 20) if with_rob else (8 << 20), names=names)      # many buckets
    batch = tu.batch_to_device(synthetic.make_batch(16, T=80, R=100, seed=5), dev)
    out = {"backend": dist.get_backend(), "with_roberta": with_rob, "gradient_bytes": int(flat.flat.numel() * 4), "buckets": len(flat.buckets), "bucket_bytes": [int((e - s_) * 4) for s_, e, _ in flat.buckets]}
    grads = []
    for forced in (True, False):
        mh.DROPOUT.manual_seed(77)
        flat.zero()
        loss = model(**tu.forward_inputs(batch))[0]
        flat.begin(1, force=forced)
        loss.backward()
        launched = flat.launched_in_backward
        flat.finish(1)
        torch.cuda.synchronize()
        grads.append(flat.flat.clone())
        if forced:
            out["launched_during_backward"] = launched
    diff = float((grads[0] - grads[1]).abs().max())
    scale = float(grads[1].abs().max())
    out.update(max_abs_diff=diff, grad_scale=scale, loss=float(loss.item()))
    # round 6: the opt-in bf16 buckets (FlatGrads(comm_dtype=)) over the same backend -- the identity collective returns every bucket
    # rounded to bf16 once, so the result must be the fp32 gradient to half a bf16 ulp (2^-8) of each element's magnitude
    flat16 = tu.FlatGrads([pd[k] for k in names], dev, bucket_bytes=(64 << 20) if with_rob else (8 << 20), names=names, comm_dtype=torch.bfloat16)
    mh.DROPOUT.manual_seed(77)
    loss = model(**tu.forward_inputs(batch))[0]
    flat16.begin(1, force=True)
    loss.backward()
    flat16.finish(1)
    torch.cuda.synchronize()
    # (per element: bf16's half ulp of its magnitude + the float-atomic noise of two runs of the same step)
    rel16 = float(((flat16.flat - grads[1]).abs() / (grads[1].abs() + 2e-3 * scale)).max())
    out.update(bf16_buckets_max_rel_diff=rel16, bf16_buckets_launched_during_backward=flat16.launched_in_backward)
    flat16.close()
    flat.install()
    for p_, k in ((pd[k], k) for k in names):           # (p.grad views belong to the buffer built last: hand them back to `flat`)
        off = flat.offsets[id(p_)]
        p_.grad = flat.flat[off:off + p_.numel()].view_as(p_)
    # and a whole optimisation step through the same path
    opt = tu.FlatAdamW(flat, names, t_total=100)
    flat.zero()
    loss = model(**tu.forward_inputs(batch))[0]
    flat.begin(1, force=True)
    loss.backward()
    flat.finish(1)
    opt.step(1.0)
    torch.cuda.synchronize()
    out["step_loss"] = float(loss.item())
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)
    # float atomics in the heads' LayerNorm / bias gradients: not bit-equal between two runs of the same step
    ok = out["backend"] == "nccl" and out["launched_during_backward"] >= 1 and diff <= 1e-5 * max(1.0, scale) and all(map(lambda v: v == v, (diff, out["step_loss"])))
    ok = ok and rel16 <= 2.0 ** -8 * 1.05 and out["bf16_buckets_launched_during_backward"] >= 1
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
