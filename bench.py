#!/usr/bin/env python
"""bench.py -- ModCR hot path on MI355X: PMR training examples/s (4-choice, S = 180).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one optimisation step of the ModCR training loop (run_PMR_ModCR.py:188-227) on one
synthetic PMR batch per GPU: image-only global_enc pass + global_enc + seq_enc (12 Oscar-base layers
each, frozen, no_grad) + multi-view alignment layers + mapping networks + scorer + 4-way CE,
backward through every trainable head, (N>1: one RCCL all-reduce of the flat gradient buffer),
grad-norm clip, AdamW step.  The 24-layer prefix RoBERTa body is SURVEY 8(f) rank 1 ("next") and is
represented by a small trainable pooler (modeling/roberta_prefix.py); config.workload says so.

Prints ONE JSON line (rank 0): metric/value/unit per BASELINE.json + `roofline` for the fused
QKV+attention forward kernel (HIP-event timed inside the timed region) + `cpu_baseline` (the CPU
oracle on a bounded sample of the same workload, rank 0, N = 1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "multimodal-context-reasoning_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_BF16 = 2.5e15       # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
B_PER_GPU = 64           # examples per GPU per step (BASELINE.json configs[1]: batch=64 -> 256 sequences)
T_TEXT, R_IMG = 80, 100  # S = 180


class KernelTimer(object):
    """HIP-event pairs around every launch of one C-ABI entry point, recorded on the stream the
    kernel is launched on (torch's current stream = the stream handed to the C ABI)."""

    def __init__(self, mh, name, select):
        self.mh, self.name, self.select = mh, name, select
        self.orig = getattr(mh, name)
        self.pairs = []
        self.enabled = False

    def __enter__(self):
        def wrapped(*a, **k):
            if not (self.enabled and self.select(*a, **k)):
                return self.orig(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            out = self.orig(*a, **k)
            e1.record()
            self.pairs.append((e0, e1))
            return out
        setattr(self.mh, self.name, wrapped)
        return self

    def __exit__(self, *exc):
        setattr(self.mh, self.name, self.orig)

    def mean_seconds(self):
        if not self.pairs:
            return None
        return sum(a.elapsed_time(b) for a, b in self.pairs) / len(self.pairs) * 1e-3


def cpu_baseline(model, seed, num_threads):
    """The CPU oracle (oracle/modcr_oracle.py, kind 'port') on a bounded sample of the same
    workload: B = 2 examples (8 sequences, T=80, R=100), same weights, forward + head backward."""
    from Data import synthetic
    from oracle import modcr_oracle as O
    torch.set_num_threads(num_threads)
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    for k, v in sd.items():
        if not (k.startswith("calec.global_enc.") or k.startswith("calec.seq_enc.")):
            v.requires_grad_(True)
    cfg = dict(hidden_size=768, num_attention_heads=12, num_hidden_layers=12, layer_norm_eps=1e-12,
               img_layer_norm_eps=1e-12, use_img_layernorm=1)
    batch = synthetic.make_batch(2, T=T_TEXT, R=R_IMG, seed=seed)
    batch["roberta_input_ids"] = batch["r_input_ids"]

    def roberta_fn(ids, tt, m, prefix_emb, prompt_mask):
        return torch.tanh(torch.nn.functional.linear(prefix_emb.reshape(prefix_emb.shape[0], -1),
                                                     sd["roberta.dense.weight"], sd["roberta.dense.bias"]))

    def one():
        for v in sd.values():
            v.grad = None
        loss, _, logits, _ = O.abstract_specific(sd, cfg, batch, roberta_fn)
        loss.backward()
        return loss, logits

    t0 = time.perf_counter()
    loss, logits = one()                    # warm-up (also the measurement if the host is slow)
    dt = time.perf_counter() - t0
    iters = 0 if dt > 12.0 else max(1, min(12, int(12.0 / max(dt, 1e-3))))      # about 10-12 s of CPU work
    if iters:
        t0 = time.perf_counter()
        for _ in range(iters):
            loss, logits = one()
        dt = (time.perf_counter() - t0) / iters
    return {"value": round(2.0 / dt, 4), "unit": "examples/s", "cores": num_threads, "kind": "port",
            "sample": "oracle/modcr_oracle.py fp32, same weights, B=2 examples (8 seq, S=180), fwd + head bwd, "
                      "%d timed iterations after 1 warm-up, %.2f s each" % (iters, dt)}, batch, loss.detach(), logits.detach()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=B_PER_GPU, help="examples per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dropout", type=float, default=0.3,
                    help="hidden_dropout_prob (the reference trains with --drop_out 0.3, live inside the frozen encoders too: "
                         "run_PMR_ModCR.py:171,585); 0 = the eval-mode arithmetic")
    ap.add_argument("--attn-dropout", type=float, default=0.1,
                    help="attention_probs_dropout_prob of the two Oscar encoders (0.1 in the BERT / Oscar checkpoints' config.json, "
                         "live in training mode; also the prefix RoBERTa body's with --with-roberta)")
    ap.add_argument("--h2d", choices=("none", "sync", "overlap"), default="none",
                    help="PCIe-inclusive variant (NOT the contract's `value`, which has inputs resident in HBM): every step's batch "
                         "starts in pinned host memory; 'sync' copies it before the step, 'overlap' copies batch i+1 on a side stream "
                         "while step i runs (what the run scripts' loader does)")
    ap.add_argument("--with-roberta", action="store_true",
                    help="include the 24-layer prefix RoBERTa-large body, forward and backward (SURVEY 8f-1); "
                         "not the default workload (BASELINE north_star names the Oscar/ChunkAlign path)")
    ap.add_argument("--train-encoders", action="store_true",
                    help="run global_enc and seq_enc WITH gradients (SURVEY 8f-4, the ChunkAlign_CLS_enc4_align variant): "
                         "every encoder layer's backward on the HIP kernels; the reference's ModCR step keeps them frozen")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch N>1 with torch.distributed.run" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://")      # nccl == RCCL on ROCm

    import modcr_hip as mh
    from Data import synthetic
    from modeling import train_utils as tu
    mh.lib()                                # fail loudly if the HIP library is not built

    model = tu.build_model(dev, seed=0, roberta_body="large" if args.with_roberta else "standin",
                           hidden_dropout_prob=args.dropout, train_encoders=args.train_encoders,
                           attention_probs_dropout_prob=args.attn_dropout, roberta_hidden_dropout_prob=0.1 if args.with_roberta else 0.0)
    mh.DROPOUT.manual_seed(1000 + rank)     # same seed on every rank = same initial weights
    model.train()
    names = tu.trainable_parameters(model)
    pdict = dict(model.named_parameters())
    for k, p in pdict.items():
        p.requires_grad_(k in names)
    params = [pdict[k] for k in names]
    flat = tu.FlatGrads(params, dev)
    if os.environ.get("MODCR_TORCH_OPTIM"):     # A/B knob: torch.optim.AdamW + clip_grad_norm_ (foreach kernels)
        opt, sched = tu.make_optimizer(model, names, t_total=100000)
    else:                                       # fused clip + AdamW over the flat buffers
        opt, sched = tu.FlatAdamW(flat, names, t_total=100000), None

    # synthetic PMR batches, resident in HBM before the timed region (different data per rank/step)
    nb = min(4, args.steps + args.warmup)
    batches = [tu.batch_to_device(synthetic.make_batch(args.batch, T=T_TEXT, R=R_IMG, seed=1234 + 97 * rank + i), dev)
               for i in range(nb)]
    n_seq = args.batch * 4
    s_len = T_TEXT + R_IMG
    host_batches, copy_stream = None, None
    if args.h2d != "none":
        def pin(v):
            return v.pin_memory() if torch.is_tensor(v) else ([t.pin_memory() for t in v] if isinstance(v, list) and v and torch.is_tensor(v[0]) else v)
        host_batches = [{k: pin(v.cpu() if torch.is_tensor(v) else ([t.cpu() for t in v] if isinstance(v, list) and v and torch.is_tensor(v[0]) else v))
                         for k, v in b.items()} for b in batches]
        copy_stream = torch.cuda.Stream()

    def fetch(i):
        """batch of step i: resident in HBM (the contract), or copied from pinned host memory"""
        if host_batches is None:
            return batches[i % nb]
        if args.h2d == "sync":
            b = tu.batch_to_device(host_batches[i % nb], dev)
            torch.cuda.current_stream().synchronize()
            return b
        main = torch.cuda.current_stream()
        if fetch.nxt is None:
            with torch.cuda.stream(copy_stream):
                fetch.nxt = tu.batch_to_device(host_batches[i % nb], dev)
        main.wait_stream(copy_stream)
        cur = fetch.nxt
        for v in cur.values():
            for t in (v if isinstance(v, list) else [v]):
                if torch.is_tensor(t):
                    t.record_stream(main)
        with torch.cuda.stream(copy_stream):
            fetch.nxt = tu.batch_to_device(host_batches[(i + 1) % nb], dev)
        return cur
    fetch.nxt = None

    def is_c2_attention(x, *a, **k):
        # the dominant kernel: qkv_attn4_kernel<1> = production call with the broadcast key mask (global_enc and the
        # phase-2 layers of seq_enc, 18 of the 24 S=180 launches per step); the dense-mask / chunk-mean / align-map
        # variants are other kernels (qkv_attn4_kernel<2>, <3>) with their own rows in the rocprof summary
        return (x.shape[1] == s_len and k.get("mask_bits") is None and k.get("chunk_id") is None
                and k.get("align_map") is None and k.get("hist") is None and not k.get("want_probs"))

    with KernelTimer(mh, "qkv_attn", is_c2_attention) as kt:
        for i in range(args.warmup):
            tu.train_step(model, fetch(i), opt, sched, flat, world)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        kt.enabled = True
        t0 = time.perf_counter()
        for i in range(args.steps):
            loss, logits = tu.train_step(model, fetch(args.warmup + i), opt, sched, flat, world)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        kt.enabled = False
        t_attn = kt.mean_seconds()
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # counters of the variant the step launches: training mode with the dropout masking, or the eval-mode kernel
    pmc_file = "r01_attn4_drop_pmc.txt" if args.attn_dropout > 0 else "r01_attn4_pmc.txt"

    def pmc_traffic():
        """HBM-side bytes per launch of the roofline kernel from the committed rocprofv3 --pmc passes
        (profiles/r01_attn4_pmc.txt: FETCH_SIZE, WRITE_SIZE in KB; FETCH_SIZE doubled as the gfx950 guide
        prescribes).  Separate passes, same shape (N=256, S=180, H=768); None when the shape differs."""
        if n_seq != 256 or s_len != 180:
            return None
        try:
            vals = {}
            for line in open(os.path.join(ROOT, "profiles", pmc_file)):
                f = line.split()
                mean = [t for t in f if t.startswith("mean=")]
                if f and f[0] in ("FETCH_SIZE", "WRITE_SIZE") and mean:
                    vals[f[0]] = float(mean[0].split("=")[1])
            return round((2.0 * vals["FETCH_SIZE"] + vals["WRITE_SIZE"]) * 1024.0)
        except Exception:
            return None

    def eval_mode_kernel_seconds(launches=20):
        """the same kernel without the training-mode dropout masking (what BASELINE config 2, 'fused prefix-attention fwd
        only', names): a few launches of the C-ABI call after the timed region, layer-0 weights of global_enc"""
        att = model.calec.global_enc.encoder.layer[0].attention.self
        w, b = att.packed_qkv(torch.bfloat16)
        x = torch.randn(n_seq, s_len, 768, device=dev).to(torch.bfloat16)
        km = batches[0]["input_mask"].to(torch.float32) if batches[0]["input_mask"].shape == (n_seq, s_len) else torch.ones(n_seq, s_len, device=dev)
        for _ in range(3):
            mh.qkv_attn(x, w, b, key_mask=km, num_heads=12)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            mh.qkv_attn(x, w, b, key_mask=km, num_heads=12)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / launches * 1e-3

    if rank == 0:
        h, a = 768, 12
        flops_attn = n_seq * (6.0 * s_len * h * h + 4.0 * s_len * s_len * h)     # SURVEY 8(d), padding not counted
        achieved = flops_attn / t_attn / 1e12
        attn_drop = args.attn_dropout > 0
        t_eval = eval_mode_kernel_seconds() if attn_drop else None
        out = {
            "metric": "PMR training examples/sec (4-choice, seq~180)",
            "value": round(args.batch * world * args.steps / elapsed, 3),
            "unit": "examples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "PMR 4-choice T=80 R=100 (S=180) H=768, %d examples (=%d sequences)/GPU/step: "
                                   "%s, "
                                   "cls_layer_lyx x2 + mapping networks + scorer + MC-CE fwd+bwd, grad clip + AdamW (fused flat-buffer step); "
                                   "%s; %s" % (args.batch, n_seq,
                                                        "Oscar-base global_enc (full S=180) + seq_enc fwd+BWD with gradients (--train-encoders, SURVEY 8f-4), "
                                                        "image-only global_enc pass S=101 fwd" if args.train_encoders else
                                                        "frozen Oscar-base global_enc (image-only S=101 + full S=180) + seq_enc fwd",
                                                        "prefix RoBERTa-large body INCLUDED (24 layers, H=1024, S=106, fwd+bwd, trainable)"
                                                        if args.with_roberta else
                                                        "prefix-RoBERTa body NOT included (stand-in pooler; SURVEY 8f-1 'next', --with-roberta adds it)",
                                                        ("hidden dropout %.2g live (embeddings, BertSelfOutput, BertOutput, heads, cross_attention_lyx weights 0.1; "
                                                         "counter-based masks), attention-probability dropout %.2g live" % (args.dropout, args.attn_dropout)) if args.dropout > 0 else "dropout off"),
                       "global_batch": args.batch * world, "seq_len": s_len, "parallelism": "dp%d" % world,
                       "inputs": {"none": "resident in HBM before the timed region", "sync": "PCIe-INCLUSIVE: copied from pinned host memory before every step",
                                  "overlap": "PCIe-INCLUSIVE: copied from pinned host memory on a side stream under the previous step"}[args.h2d]},
            "roofline": {"kernel": "qkv_attn4_kernel<1,192,%d> (fused QKV projection + attention fwd%s, N=%d S=%d H=%d)"
                                   % (1 if attn_drop else 0, ", training mode: attention-probability dropout mask applied in the kernel"
                                      if attn_drop else "", n_seq, s_len, h),
                         "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16 / 1e12,
                         "unit": "TFLOP/s", "frac": round(achieved * 1e12 / PEAK_BF16, 4),
                         "launches_timed": len(kt.pairs), "avg_launch_us": round(t_attn * 1e6, 2),
                         "algorithmic_gflop_per_launch": round(flops_attn / 1e9, 2), "traffic": pmc_traffic(),
                         "traffic_unit": "bytes per launch (2*FETCH_SIZE + WRITE_SIZE, profiles/%s)" % pmc_file},
            "loss": round(float(loss.item()), 5),
        }
        if t_eval:      # informational: the eval-mode variant of the same kernel (no dropout masking), outside the timed region
            out["roofline"]["eval_mode_variant"] = {"kernel": "qkv_attn4_kernel<1,192,0>", "avg_launch_us": round(t_eval * 1e6, 2),
                                                    "achieved": round(flops_attn / t_eval / 1e12, 2),
                                                    "frac": round(flops_attn / t_eval / PEAK_BF16, 4),
                                                    "launches_timed": 20, "where": "after the timed region, same shape and mask"}
        if world == 1 and not args.no_cpu_baseline and not args.with_roberta and not args.train_encoders:
            try:
                ncpu = len(os.sched_getaffinity(0))
            except AttributeError:
                ncpu = os.cpu_count() or 1
            base, cbatch, closs, clogits = cpu_baseline(model, 4321, max(1, min(32, ncpu)))
            out["cpu_baseline"] = base
            # live parity of the same sample through the HIP path (eval-free: dropout is off)
            model.eval()                         # the oracle has no dropout: compare the eval-mode arithmetic
            with torch.no_grad():
                o = model(**tu.forward_inputs(tu.batch_to_device(cbatch, dev)))
            model.train()
            out["parity_vs_oracle"] = {"max_abs_logit_err": round(float((o[2].float().cpu() - clogits).abs().max()), 5),
                                       "loss_err": round(abs(float(o[0].item()) - float(closs)), 6),
                                       "argmax_agree": bool((o[2].argmax(-1).cpu() == clogits.argmax(-1)).all())}
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
