"""Trainable-encoder step at the bench shape: finite-gradient report; STEPS=n runs optimizer steps, REPEAT=n repeats one
step with the same dropout counters and lists parameters whose gradient is not reproducible."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import modcr_hip as mh
from Data import synthetic
from modeling import train_utils as tu
dev = torch.device("cuda")
p = float(os.environ.get("P", "0.3"))
model = tu.build_model(dev, seed=0, hidden_dropout_prob=p, train_encoders=True)
model.train()
names = tu.trainable_parameters(model)
pd = dict(model.named_parameters())
for k, q in pd.items():
    q.requires_grad_(k in names)
b = tu.batch_to_device(synthetic.make_batch(int(os.environ.get("B", "64")), T=80, R=100, seed=1234), dev)
out = model(**tu.forward_inputs(b))
print("loss", float(out[0].item()))
out[0].backward()
bad = [k for k in names if pd[k].grad is None or not torch.isfinite(pd[k].grad).all()]
print("non-finite grads:", len(bad), bad[:12])
for k in names:
    if pd[k].grad is not None and "layer.11.output.dense.weight" in k:
        print(k, float(pd[k].grad.norm()))
tot = torch.sqrt(sum((pd[k].grad.float() ** 2).sum() for k in names if pd[k].grad is not None))
print("grad norm", float(tot))
if os.environ.get("STEPS"):
    for q in pd.values():
        q.grad = None
    params = [pd[k] for k in names]
    flat = tu.FlatGrads(params, dev)
    opt = tu.FlatAdamW(flat, names, t_total=100000)
    for it in range(int(os.environ["STEPS"])):
        bb = tu.batch_to_device(synthetic.make_batch(int(os.environ.get("B", "64")), T=80, R=100, seed=1234 + it), dev)
        out = model(**tu.forward_inputs(bb))
        flat.begin(1)
        out[0].backward()
        flat.finish(1)
        badg = [k for k in names if not torch.isfinite(pd[k].grad).all()]
        print("step", it, "loss", float(out[0].item()), "gradnorm", float(flat.flat.norm()), "bad grads", len(badg), badg[:6], flush=True)
        top = sorted(((float(pd[k].grad.abs().max()), k) for k in names), reverse=True)[:8]
        print("   top", top, flush=True)
        opt.step(1.0)
        flat.zero()
        badp = [k for k in names if not torch.isfinite(pd[k]).all()]
        print("   bad params", len(badp), badp[:6], flush=True)
        if badg or badp:
            break
if os.environ.get("REPEAT"):
    for q in pd.values():
        q.grad = None
    bb = tu.batch_to_device(synthetic.make_batch(int(os.environ.get("B", "64")), T=80, R=100, seed=1235), dev)
    ref = None
    for it in range(int(os.environ["REPEAT"])):
        mh.DROPOUT.manual_seed(5)
        for q in pd.values():
            q.grad = None
        out = model(**tu.forward_inputs(bb))
        out[0].backward()
        cur = {k: pd[k].grad.detach().clone() for k in names}
        if ref is None:
            ref = cur
            continue
        worst = []
        for k in names:
            d = float((cur[k] - ref[k]).abs().max())
            sc = float(ref[k].abs().max()) + 1e-12
            if not (d <= 0.05 * sc):
                worst.append((d / sc, d, k))
        worst.sort(reverse=True)
        print("repeat", it, "loss", float(out[0].item()), "params differing >5%:", len(worst), worst[:6], flush=True)
