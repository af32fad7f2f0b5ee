#!/usr/bin/env python
"""Feasibility / timing of the headline step under a hipGraph (torch.cuda.CUDAGraph): forward + backward of bench.py's step captured
once and replayed, the optimizer step launched eagerly behind it.  TIMING ONLY in this form: the dropout counters and the loss scale
are kernel arguments, so a replay repeats the captured masks.  usage: graph_step.py [BATCH=128] [STEPS=20] [MODE=fwdbwd|all]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh  # noqa: E402
from Data import synthetic  # noqa: E402
from modeling import train_utils as tu  # noqa: E402

batch_n, steps = int(os.environ.get("BATCH", 128)), int(os.environ.get("STEPS", 20))
train_enc = bool(int(os.environ.get("TRAIN_ENC", 0)))
dev = torch.device("cuda")
mh.lib()
model = tu.build_model(dev, seed=0, roberta_body="standin", hidden_dropout_prob=0.3, train_encoders=train_enc, attention_probs_dropout_prob=0.1,
                       hidden_size=768, num_hidden_layers=12, num_attention_heads=12)
model.train()
names = tu.trainable_parameters(model)
pd = dict(model.named_parameters())
for k, p in pd.items():
    p.requires_grad_(k in names)
flat = tu.FlatGrads([pd[k] for k in names], dev, names=names)
opt = tu.FlatAdamW(flat, names, t_total=100000)
mh.DROPOUT.manual_seed(1000)
batch = tu.batch_to_device(synthetic.make_batch(batch_n, T=80, R=100, seed=1234), dev)
inputs = tu.forward_inputs(batch)


def fwd_bwd():
    loss = model(**inputs)[0]
    flat.begin(1)
    loss.backward()
    flat.finish(1)
    return loss


def timed(fn, n):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def eager_step():
    fwd_bwd()
    opt.step(1.0)
    flat.zero()


for _ in range(3):
    eager_step()
print("eager: %.3f ms per step" % timed(eager_step, steps), flush=True)

side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(2):
        eager_step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        static_loss = fwd_bwd()
except Exception as e:                  # what breaks the capture is the result of this experiment too
    print("capture failed: %s: %s" % (type(e).__name__, str(e)[:600]), flush=True)
    sys.exit(1)
torch.cuda.synchronize()


def graph_step():
    g.replay()
    opt.step(1.0)
    flat.zero()


graph_step()
print("graph:  %.3f ms per step (forward + backward replayed, optimizer eager), loss %.5f" % (timed(graph_step, steps), float(static_loss)), flush=True)
print("eager again: %.3f ms per step" % timed(eager_step, steps), flush=True)
