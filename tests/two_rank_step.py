#!/usr/bin/env python
"""Helper of tests/test_hip_models.py::test_two_ranks_equal_one_rank_on_the_full_batch (not a test
module itself).  Run as `python -m torch.distributed.run --nproc-per-node 2 tests/two_rank_step.py --plan heads|encoders
--out DIR`, both ranks on device 0, collectives over gloo (RCCL refuses two ranks on one device): every rank builds the same
model (seed 0), takes ITS half of one synthetic batch of 4 examples (the four choices of an example stay together), runs
K steps of the product loop (modeling/train_utils.py::micro_step with world_size 2: bucketed all-reduce launched from the
gradient sink's reports and autograd's hooks while the HIP backward runs, 1 / world scaling, clip on the reduced gradient,
AdamW) and saves the first step's reduced gradient and the final flat parameter buffer to DIR/rank<r>.pt.
With --world1 the same script runs as ONE process on the full batch (the reference result)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "multimodal-context-reasoning_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def slice_examples(batch, lo, hi, choices=4):
    import torch
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v):
            out[k] = v[lo * choices:hi * choices].contiguous()
        elif isinstance(v, list):
            out[k] = v[lo * choices:hi * choices]
        else:
            out[k] = v
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plan", choices=["heads", "encoders"], required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--world1", action="store_true")
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    world = 1 if args.world1 else int(os.environ["WORLD_SIZE"])
    rank = 0 if args.world1 else int(os.environ["RANK"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", init_method="env://")
    import modcr_hip as mh
    from Data import synthetic
    from modeling import train_utils as tu
    mh.lib()
    dims = dict(hidden_size=768, num_hidden_layers=12, num_attention_heads=12) if args.plan == "heads" else \
        dict(hidden_size=256, num_hidden_layers=12, num_attention_heads=4)
    model = tu.build_model(dev, seed=0, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0, train_encoders=args.plan == "encoders",
                           vocab_size=3000, max_position_embeddings=128, img_feature_dim=70, **dims)
    model.eval()        # dropout off everywhere (the heads' Dropout(0.1) modules are hard-coded, modeling_ensemble.py:439-457, v10:780)
    names = tu.trainable_parameters(model)
    pd = dict(model.named_parameters())
    for k, p in pd.items():
        p.requires_grad_(k in names)
    # several buckets, so that some are launched from inside backward
    flat = tu.FlatGrads([pd[k] for k in names], dev, bucket_bytes=(32 << 20) if args.plan == "heads" else (1 << 20), names=names)
    opt = tu.FlatAdamW(flat, names, 1e-4, adam_epsilon=1e-5, t_total=20)
    per = 4 // world
    grad0, launched = None, []
    for t in range(args.steps):
        full = synthetic.make_batch(4, T=80, R=50, seed=70 + t, vocab_size=3000, img_dim=70, roberta_len=16)
        b = tu.batch_to_device(slice_examples(full, rank * per, (rank + 1) * per), dev)
        if t == 0:
            # first step by hand, to keep the reduced gradient: forward, begin, backward, finish = micro_step's first half
            loss = model(**tu.forward_inputs(b))[0]
            flat.begin(world)
            loss.backward()
            launched.append(flat.launched_in_backward)
            flat.finish(world)
            grad0 = flat.flat.detach().clone()
            opt.step(1.0)
            flat.zero()
        else:
            tu.micro_step(model, b, opt, None, flat, world, 1.0, 1, True)
            launched.append(flat.launched_in_backward)
    torch.cuda.synchronize()
    os.makedirs(args.out, exist_ok=True)
    torch.save({"grad0": grad0.cpu(), "params": opt.flat_p.detach().cpu(), "buckets": len(flat.buckets), "launched": launched,
                "layout": list(opt.layout)}, os.path.join(args.out, "world%d_rank%d.pt" % (world, rank)))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
