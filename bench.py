#!/usr/bin/env python
"""bench.py -- ModCR hot path on MI355X: PMR training examples/s (4-choice, S = 180).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

`python bench.py --gpus N` without a launcher starts the N ranks itself (child processes, one per GPU, started BEFORE
anything touches the GPU) -- both command shapes give the same JSON line.

One step = one optimisation step of the ModCR training loop (run_PMR_ModCR.py:188-227) on one synthetic PMR batch of
128 examples (= 512 sequences, BASELINE configs[2] / [3]) per GPU: image-only global_enc pass + global_enc + seq_enc
(12 Oscar-base layers each, frozen, no_grad, dropouts live as under model.train()) + multi-view alignment layers +
mapping networks + scorer + 4-way CE, backward through every trainable head, (N>1: bucketed RCCL all-reduce of the flat
gradient buffer overlapped with backward), grad-norm clip, transformers-AdamW step.  The 24-layer prefix RoBERTa body is
SURVEY 8(f) rank 1 ("next") and is represented by a small trainable pooler (modeling/roberta_prefix.py); config.workload
says so; --with-roberta includes it.

Prints ONE JSON line (rank 0): metric/value/unit per BASELINE.json +
  roofline      fused QKV+attention forward kernel, HIP-event timed inside the timed region (the shape the step runs:
                512 sequences) and, as `config2`, at BASELINE configs[1] (N = 256 sequences, fwd only, eval and train variants)
  cpu_baseline  the CPU oracle (kind "port") on bounded samples of the same workload: full step (B = 2), and SURVEY 8(d)'s
                cases (i) fused-attention fwd and (ii) one layer fwd+bwd
  parity_vs_oracle  answer-agreement RATE of the HIP path against the oracle over synthetic "val" examples, margin-aware
  config3_full_fwd_bwd  the same step with both encoders trained (every layer's backward on the HIP kernels)
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "multimodal-context-reasoning_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_BF16 = 2.5e15       # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
B_PER_GPU = 128          # examples per GPU per step (BASELINE.json configs[2]/[3]: batch=128 -> 512 sequences; 8 GPUs -> global 1024)
T_TEXT, R_IMG = 80, 100  # S = 180
H_OSCAR, A_OSCAR, L_OSCAR = 768, 12, 12
# --config c5: the shape class of BASELINE.json configs[4] (run_vcr_ModCR.py, Oscar-large): H = 1024, 16 heads, 24 layers, T = 194 text tokens +
# R = 36 regions = S 230.  Not a configuration the reference itself can build (its mapping networks hard-code 768-wide inputs and its seq_enc
# schedule 12 layers): this build widens the heads with the encoder and scales the phase schedule (first quarter / middle
# half / last quarter of the layers).
CONFIGS = {"pmr": dict(T=80, R=100, H=768, A=12, L=12, batch=128),
           "c5": dict(T=194, R=36, H=1024, A=16, L=24, batch=32),
           # toy dims: NOT a measurement -- the many-rank rehearsal of the N > 1 plumbing (spawn, port, weight broadcast, bucket count-down,
           # per-rank timing) on a one-GPU box, where every rank's model must be built in seconds (tests/test_hip_models.py)
           "toy": dict(T=24, R=12, H=256, A=4, L=4, batch=2)}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=sorted(CONFIGS), default="pmr",
                    help="pmr = the headline PMR workload (BASELINE configs[1..3]); c5 = the VCR / Oscar-large shape class of configs[4]; "
                         "toy = small dims for the many-rank rehearsal of the N > 1 plumbing (not a measurement)")
    ap.add_argument("--batch", type=int, default=None, help="examples per GPU per step (default: 128 for pmr, 32 for c5 = VCR's 8 x 4 accumulation)")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip cpu_baseline and parity_vs_oracle (CPU work)")
    ap.add_argument("--parity-examples", type=int, default=256,
                    help="synthetic 'val' examples for the answer-agreement rate against the CPU oracle (time-boxed)")
    ap.add_argument("--parity-seconds", type=float, default=120.0, help="time box of the oracle side of the agreement check")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="REHEARSAL of the N > 1 path on a one-GPU box: every rank uses device 0 and the collectives run over gloo "
                         "(RCCL refuses two ranks on one device).  Exercises spawn, weight broadcast, the bucketed all-reduce hooks "
                         "and the max-over-ranks timing; the printed value is NOT a measurement (config.rehearsal says so)")
    ap.add_argument("--no-config3", action="store_true", help="skip the second measurement (both encoders trained)")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the secondary workloads of the default line (with_roberta = the reference's real step with the prefix "
                         "RoBERTa-large body trained; c5 = the VCR / Oscar-large shape class)")
    ap.add_argument("--real-step-seconds", type=float, default=40.0,
                    help="time box of the CPU-oracle evidence of the with_roberta leg (its own cpu_baseline + agreement check)")
    ap.add_argument("--leg-seconds", type=float, default=60.0, help="time box of each secondary workload (model build excluded)")
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
    ap.add_argument("--last-layer-rows", action="store_true",
                    help="opt-in config.modcr_last_layer_rows for the MAIN measurement: the frozen encoders' last layers skip the token-wise blocks of rows "
                         "ModCR never reads (same loss, logits and gradients).  Off by default: the headline computes every row the reference computes; "
                         "the default line reports the opt-in step as the secondary field `last_layer_rows`")
    ap.add_argument("--with-roberta", action="store_true",
                    help="include the 24-layer prefix RoBERTa-large body, forward and backward (SURVEY 8f-1); "
                         "not the default workload (BASELINE north_star names the Oscar/ChunkAlign path)")
    ap.add_argument("--train-encoders", action="store_true",
                    help="run global_enc and seq_enc WITH gradients as the headline (SURVEY 8f-4, the ChunkAlign_CLS_enc4_align "
                         "variant); by default this is measured second and reported as config3_full_fwd_bwd")
    ap.add_argument("--optimizer", choices=("hf", "torch"), default="hf",
                    help="hf = transformers.AdamW arithmetic, what the reference trains with (default); torch = torch.optim.AdamW form (A/B)")
    ap.add_argument("--grad-comm", choices=("fp32", "bf16"), default="fp32",
                    help="N > 1: the type the gradient buckets cross the links in (fp32 = the default and the contract's `value`; bf16 = the "
                         "opt-in half-size buckets of FlatGrads(comm_dtype=), recorded in config.gradient_buckets)")
    ap.add_argument("--allow-knobs", action="store_true", help="run although MODCR_* environment variables are set (they are recorded)")
    return ap.parse_args()


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(args, cmd=None):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (this process never touches
    the GPU and never execs) and exit with the first non-zero code.  cmd: the child command (tests)."""
    port = free_port()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen(cmd or ([sys.executable, os.path.abspath(__file__)] + sys.argv[1:]), env=env))
    # poll all of them: the first rank that dies (OOM, missing .so, RCCL init failure) takes the others down at once -- they
    # would otherwise sit in the rendezvous or in a collective until its timeout and the harness would lose the run
    rc = 0
    live = list(procs)
    while live and not rc:
        time.sleep(0.2)
        for p in list(live):
            code = p.poll()
            if code is not None:
                live.remove(p)
                rc = rc or code
    if rc:
        for p in live:
            p.terminate()
        deadline = time.time() + 10.0
        for p in live:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc


def hip_runtime():
    """the HIP runtime this process already uses (torch's), by its mapped path -- a second copy would not know our streams"""
    import ctypes
    path = None
    for line in open("/proc/self/maps"):
        if "libamdhip64" in line:
            path = line.split()[-1]
            break
    if path is None:
        raise RuntimeError("libamdhip64 is not loaded")
    hip = ctypes.CDLL(path)
    hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
    hip.hipEventSynchronize.argtypes = [ctypes.c_void_p]
    hip.hipEventElapsedTime.argtypes = [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p, ctypes.c_void_p]
    hip.hipEventDestroy.argtypes = [ctypes.c_void_p]
    return hip


class KernelTimer(object):
    """HIP events stamped by the kernel launch itself, for every selected launch of one C-ABI entry point: the entry's
    library call takes a (start, stop) hipEvent_t pair (modcr_time_next_attn) and launches with hipExtLaunchKernel on the
    stream it is handed (torch's current stream), so the pair brackets exactly the kernel -- the duration rocprofv3
    --kernel-trace reports.  (hipEventRecord before / after the call measured 5 % more on a 370 us launch inside the step:
    two extra barrier packets and their dispatch latency.)"""

    def __init__(self, mh, name, select):
        import ctypes
        self.mh, self.name, self.select, self.ct = mh, name, select, ctypes
        self.orig = getattr(mh, name)
        self.pairs = []                         # (start, stop) of every timed launch whose select() returned True / the default label
        self.labelled = {}                      # label -> pairs, for select() functions that return a string
        self.enabled = False
        self.hip = hip_runtime()

    def _event(self):
        e = self.ct.c_void_p()
        rc = self.hip.hipEventCreate(self.ct.byref(e))
        if rc != 0:
            raise RuntimeError("hipEventCreate failed (%d)" % rc)
        return e

    def __enter__(self):
        def wrapped(*a, **k):
            lab = self.select(*a, **k) if self.enabled else None
            if not lab:
                return self.orig(*a, **k)
            e0, e1 = self._event(), self._event()
            self.mh.lib().modcr_time_next_attn(e0, e1)
            out = self.orig(*a, **k)
            (self.pairs if lab is True else self.labelled.setdefault(lab, [])).append((e0, e1))
            return out
        setattr(self.mh, self.name, wrapped)
        return self

    def __exit__(self, *exc):
        setattr(self.mh, self.name, self.orig)

    def mean_seconds(self, label=None):
        pairs = self.pairs if label is None else self.labelled.get(label, [])
        if not pairs:
            return None
        tot = 0.0
        for e0, e1 in pairs:
            self.hip.hipEventSynchronize(e1)
            ms = self.ct.c_float()
            rc = self.hip.hipEventElapsedTime(self.ct.byref(ms), e0, e1)
            if rc != 0:
                raise RuntimeError("hipEventElapsedTime failed (%d)" % rc)
            tot += ms.value
        return tot / len(pairs) * 1e-3


def oracle_state(model):
    import torch
    sd = {k: v.detach().float().cpu() for k, v in model.state_dict().items()}
    cfg = dict(hidden_size=H_OSCAR, num_attention_heads=A_OSCAR, num_hidden_layers=L_OSCAR, layer_norm_eps=1e-12,
               img_layer_norm_eps=1e-12, use_img_layernorm=1)

    def roberta_fn(ids, tt, m, prefix_emb, prompt_mask):       # the stand-in pooler of modeling/roberta_prefix.py
        return torch.tanh(torch.nn.functional.linear(prefix_emb.reshape(prefix_emb.shape[0], -1),
                                                     sd["roberta.dense.weight"], sd["roberta.dense.bias"]))
    return sd, cfg, roberta_fn


def timed(fn, budget_s, max_iters=12):
    """1 warm-up call, then about `budget_s` seconds of timed calls (at least 1, at most max_iters): (seconds per call, iterations)"""
    t0 = time.perf_counter()
    fn()
    dt = time.perf_counter() - t0
    iters = max(1, min(max_iters, int(budget_s / max(dt, 1e-4))))
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    return (time.perf_counter() - t0) / iters, iters


def usable_cpus():
    """(cores in the affinity mask, CPUs the cgroup quota allows): modeling/train_utils.py::usable_cpus"""
    from modeling import train_utils as tu
    return tu.usable_cpus()


def cpu_baseline(model, seed, num_threads, host_cores=None, affinity_cores=None):
    """The CPU oracle (oracle/modcr_oracle.py, kind 'port') on bounded samples of the same workload, SURVEY 8(d):
    (iii) the full step at B = 2 (= the headline unit, examples/s), (i) the fused-attention forward, (ii) one encoder layer
    forward + backward -- fp32, torch CPU ops on `num_threads` host threads, same weights as the GPU run."""
    import torch
    from Data import synthetic
    from oracle import modcr_oracle as O
    torch.set_num_threads(num_threads)
    sd, cfg, roberta_fn = oracle_state(model)
    for k, v in sd.items():
        if not (k.startswith("calec.global_enc.") or k.startswith("calec.seq_enc.")):
            v.requires_grad_(True)
    batch = synthetic.make_batch(2, T=T_TEXT, R=R_IMG, seed=seed)
    batch["roberta_input_ids"] = batch["r_input_ids"]

    def step():
        for v in sd.values():
            v.grad = None
        loss, _, logits, _ = O.abstract_specific(sd, cfg, batch, roberta_fn)
        loss.backward()
    dt, iters = timed(step, 10.0)
    host_cores = host_cores or num_threads
    pinfo = " | ".join(ln.strip() for ln in torch.__config__.parallel_info().splitlines() if ln.strip() and ("thread" in ln.lower() or "MKL" in ln or "OpenMP" in ln))
    out = {"value": round(2.0 / dt, 4), "unit": "examples/s", "cores": num_threads, "threads": num_threads, "host_cores": host_cores,
           "affinity_mask_cores": affinity_cores or host_cores, "kind": "port",
           "cores_is": "the torch intra-op threads the timed sample ran on (`threads`); `host_cores` = CPUs this process may use (affinity mask "
                       "capped by the cgroup's cpu.max quota); `affinity_mask_cores` = cores in the mask",
           "parallel_info": pinfo[:600],
           "sample": "oracle/modcr_oracle.py fp32, same weights, B=2 examples (8 seq, S=180): 36 Oscar-base layer forwards + head fwd/bwd, "
                     "%d timed iterations after 1 warm-up, %.2f s each" % (iters, dt)}
    if host_cores > num_threads:
        # the thread cap is a measured choice: the same B = 2 step once more on every core of the affinity mask
        torch.set_num_threads(host_cores)
        dt_all, it_all = timed(step, 5.0, max_iters=4)
        torch.set_num_threads(num_threads)
        out["all_cores"] = {"threads": host_cores, "value": round(2.0 / dt_all, 4), "unit": "examples/s",
                            "sample": "the same B=2 step on all %d cores, %d iterations, %.2f s each (a B=2 step does not scale past ~32 threads: "
                                      "the 8 x 180 x 768 GEMMs are too small)" % (host_cores, it_all, dt_all)}
        if dt_all < dt:             # report the faster setting as THE baseline
            out["value"], out["cores"], out["threads"] = round(2.0 / dt_all, 4), host_cores, host_cores
            out["capped"] = {"threads": num_threads, "value": round(2.0 / dt, 4)}
            torch.set_num_threads(host_cores)
    # (i) fused-attention forward and (ii) one layer fwd+bwd on 8 sequences of the same shape, layer 0 of global_enc
    n, s = 8, T_TEXT + R_IMG
    pre = "calec.global_enc.encoder.layer.0."
    lsd = {k: v.detach().clone() for k, v in sd.items() if k.startswith(pre)}
    x = torch.randn(n, s, H_OSCAR)
    mask = O.extend_mask(batch["input_mask"][:n])
    with torch.no_grad():
        dt_a, it_a = timed(lambda: O.self_attention(x, mask, lsd, pre + "attention.self.", A_OSCAR), 4.0)
    fl_a = n * (6.0 * s * H_OSCAR ** 2 + 4.0 * s * s * H_OSCAR)
    for v in lsd.values():
        v.requires_grad_(True)
    xg = x.clone().requires_grad_(True)

    def layer_fb():
        for v in lsd.values():
            v.grad = None
        xg.grad = None
        y, _ = O.bert_layer(xg, mask, lsd, pre, A_OSCAR, 1e-12)
        y.sum().backward()
    dt_l, it_l = timed(layer_fb, 5.0)
    fl_l = 3.0 * n * (24.0 * s * H_OSCAR ** 2 + 4.0 * s * s * H_OSCAR)
    out["attention_fwd"] = {"value": round(fl_a / dt_a / 1e9, 2), "unit": "GFLOP/s", "sequences_per_s": round(n / dt_a, 2),
                            "sample": "oracle.self_attention (QKV + softmax attention), %d sequences S=%d H=%d, %d iterations, %.3f s each" % (n, s, H_OSCAR, it_a, dt_a)}
    out["layer_fwd_bwd"] = {"value": round(fl_l / dt_l / 1e9, 2), "unit": "GFLOP/s", "sequences_per_s": round(n / dt_l, 2),
                            "sample": "oracle.bert_layer forward + autograd backward, %d sequences S=%d, %d iterations, %.3f s each" % (n, s, it_l, dt_l)}
    return out


def real_step_evidence(model, dev, num_threads, budget_s, seed=777):
    """The reference's REAL step (frozen Oscar encoders + heads + the 24-layer prefix RoBERTa-large body, run_PMR_ModCR.py:201-227) on
    the CPU oracle: (a) cpu_baseline -- forward + backward of one B = 2 step, oracle.abstract_specific with oracle.roberta_prefix as
    the body, fp32, `num_threads` host threads; (b) a bounded agreement check -- eval-mode logits of the HIP path against the oracle
    on fresh synthetic examples (margin-aware like agreement_rate).  The body's splice is this build's documented choice (the
    reference's module is absent from its tree: parity unpinned there), its weights are random-init."""
    import torch
    from Data import synthetic
    from modeling import train_utils as tu
    from oracle import modcr_oracle as O
    torch.set_num_threads(num_threads)
    sd, cfg, _ = oracle_state(model)
    rob = model.roberta
    rcfg = dict(num_hidden_layers=len(rob.encoder.layer), num_attention_heads=rob.a, layer_norm_eps=rob.eps, pad_token_id=rob.pad)

    def roberta_fn(ids, tt, m, prefix_emb, prompt_mask):
        return O.roberta_prefix(sd, "roberta.", rcfg, ids, tt, m, prefix_emb, prompt_mask)[1]
    for k, v in sd.items():
        if not (k.startswith("calec.global_enc.") or k.startswith("calec.seq_enc.")):
            v.requires_grad_(True)
    batch = synthetic.make_batch(2, T=T_TEXT, R=R_IMG, seed=seed)
    batch.update(roberta_input_ids=batch["r_input_ids"], roberta_token_type_ids=batch["r_token_type_ids"], roberta_attention_mask=batch["r_attention_mask"])

    def step():
        for v in sd.values():
            v.grad = None
        O.abstract_specific(sd, cfg, batch, roberta_fn)[0].backward()
    t0 = time.perf_counter()
    step()
    dt, iters, warm = time.perf_counter() - t0, 1, 0
    if dt < 0.25 * budget_s:            # room for a warm-up: time further iterations instead of the first call
        dt, iters = timed(step, budget_s * 0.5 - dt, max_iters=3)
        warm = 1
    cpu = {"value": round(2.0 / dt, 4), "unit": "examples/s", "cores": num_threads, "kind": "port",
           "sample": "oracle/modcr_oracle.py fp32, same weights, B=2 examples (8 sequences): 36 Oscar-base layer forwards (S=180 / 101) + heads + the "
                     "24-layer prefix RoBERTa-large body (S=106) forward AND backward, %d timed iteration(s) after %d warm-up, %.2f s each" % (iters, warm, dt)}
    for v in sd.values():
        v.requires_grad_(False)
        v.grad = None
    t0 = time.perf_counter()
    hip, ora = [], []
    model.eval()
    done, chunk = 0, 4
    while done < 32 and (time.perf_counter() - t0 < budget_s * 0.5 or done == 0):
        b = synthetic.make_batch(chunk, T=T_TEXT, R=R_IMG, seed=91001 + done)
        with torch.no_grad():
            hip.append(model(**tu.forward_inputs(tu.batch_to_device(b, dev)))[2].float().cpu())
            b.update(roberta_input_ids=b["r_input_ids"], roberta_token_type_ids=b["r_token_type_ids"], roberta_attention_mask=b["r_attention_mask"])
            ora.append(O.abstract_specific(sd, cfg, b, roberta_fn)[2])
        done += chunk
    model.train()
    hip, ora = torch.cat(hip), torch.cat(ora)
    max_err = float((hip - ora).abs().max())
    top2 = ora.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    same = hip.argmax(1) == ora.argmax(1)
    dec = margin > 2.0 * max_err
    par = {"examples": int(done), "agree": round(float(same.float().mean()), 4), "decidable": int(dec.sum()),
           "agree_where_margin_gt_2tol": (round(float(same[dec].float().mean()), 4) if int(dec.sum()) else None),
           "max_abs_logit_err": round(max_err, 6), "logit_scale": round(float(ora.abs().max()), 4),
           "err_over_scale": round(max_err / max(1.0, float(ora.abs().max())), 5), "median_margin": round(float(margin.median()), 6),
           "disagreements_with_margin_gt_2tol": int((~same & dec).sum()), "oracle_seconds": round(time.perf_counter() - t0, 1),
           "note": "eval mode, random-init weights; bf16 contract 2e-2 of max(1, |logit|); the body's prefix splice is this build's documented choice "
                   "(oracle.roberta_prefix restates it: parity unpinned against the reference's absent module)"}
    return cpu, par


def agreement_rate(model, dev, n_examples, budget_s, num_threads):
    """Answer-agreement of the HIP path with the CPU oracle over synthetic 'val' examples (eval mode, as run_PMR_ModCR.py:243-280
    computes accuracy: argmax of the [B,4] logits).  Margin-aware: with random-init heads the four logits of an example lie
    within ~1e-2 of each other, so a flip only counts as a disagreement where the oracle's top-2 margin exceeds twice the
    largest logit error observed (a smaller margin cannot be decided at the bf16 contract's accuracy)."""
    import torch
    from Data import synthetic
    from modeling import train_utils as tu
    from oracle import modcr_oracle as O
    torch.set_num_threads(num_threads)
    sd, cfg, roberta_fn = oracle_state(model)
    chunk = 16
    t0 = time.perf_counter()
    hip, ora = [], []
    model.eval()
    done = 0
    while done < n_examples and (time.perf_counter() - t0 < budget_s or done == 0):
        b = synthetic.make_batch(chunk, T=T_TEXT, R=R_IMG, seed=90001 + done)
        with torch.no_grad():
            o = model(**tu.forward_inputs(tu.batch_to_device(b, dev)))
            hip.append(o[2].float().cpu())
            b["roberta_input_ids"] = b["r_input_ids"]
            ora.append(O.abstract_specific(sd, cfg, b, roberta_fn)[2])
        done += chunk
    model.train()
    hip, ora = torch.cat(hip), torch.cat(ora)
    err = (hip - ora).abs().max(dim=1).values
    max_err = float(err.max())
    top2 = ora.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    same = hip.argmax(1) == ora.argmax(1)
    dec = margin > 2.0 * max_err
    return {"examples": int(done), "agree": round(float(same.float().mean()), 4),
            "decidable": int(dec.sum()), "agree_where_margin_gt_2tol": (round(float(same[dec].float().mean()), 4) if int(dec.sum()) else None),
            "tol": round(max_err, 6), "tol_is": "largest |logit_hip - logit_oracle| over the sample (bf16 contract: 2e-2)",
            "median_margin": round(float(margin.median()), 6), "max_abs_logit_err": round(max_err, 6),
            "disagreements_with_margin_gt_2tol": int((~same & dec).sum()),
            "oracle_seconds": round(time.perf_counter() - t0, 1),
            "note": "eval mode (dropout off), random-init weights: small top-2 margins are expected; every example whose oracle margin "
                    "exceeds 2 x tol must agree"}


def main():
    global T_TEXT, R_IMG, H_OSCAR, A_OSCAR, L_OSCAR
    args = parse_args()
    cfgc = CONFIGS[args.config]
    T_TEXT, R_IMG, H_OSCAR, A_OSCAR, L_OSCAR = cfgc["T"], cfgc["R"], cfgc["H"], cfgc["A"], cfgc["L"]
    if args.batch is None:
        args.batch = cfgc["batch"]
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))           # before anything touches the GPU
    knobs = sorted(k for k in os.environ if k.startswith("MODCR_"))
    if knobs and not args.allow_knobs:
        raise SystemExit("bench.py: refusing to run with tuning knobs in the environment (%s); the product library ignores the "
                         "csrc ones, but Python-side ones would change the workload.  --allow-knobs records them instead." % ", ".join(knobs))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    if args.rehearse_on_one_gpu:
        local_rank = 0
    if not torch.cuda.is_available() or torch.cuda.device_count() <= local_rank:
        raise SystemExit("bench.py rank %d of %d: no MI355X visible as device %d (the ModCR hot path has no CPU fallback)" % (rank, world, local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        # N ranks share the box's CPU quota (16 CPUs under the cgroup of this pool's GPU boxes, 256 in the affinity mask): torch's default
        # of one intra-op thread per visible core would put N x 256 threads on 16 CPUs for every host-side torch op of the launch loop
        from modeling import train_utils as _tu
        _tu.cap_host_threads(int(os.environ.get("LOCAL_WORLD_SIZE", str(world))))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo" if args.rehearse_on_one_gpu else "nccl", init_method="env://")      # nccl == RCCL on ROCm

    import modcr_hip as mh
    from Data import synthetic
    from modeling import train_utils as tu
    mh.lib()                                # fail loudly if the HIP library is not built
    assert not mh.is_tuning_library()

    def setup(train_encoders, with_roberta, dims=None, last_layer_rows=None):
        h_, l_, a_ = dims or (H_OSCAR, L_OSCAR, A_OSCAR)
        llr = args.last_layer_rows if last_layer_rows is None else last_layer_rows
        model = tu.build_model(dev, seed=0, roberta_body="large" if with_roberta else "standin",
                               hidden_dropout_prob=args.dropout, train_encoders=train_encoders,
                               attention_probs_dropout_prob=args.attn_dropout, roberta_hidden_dropout_prob=0.1 if with_roberta else 0.0,
                               hidden_size=h_, num_hidden_layers=l_, num_attention_heads=a_, **({"modcr_last_layer_rows": True} if llr else {}))
        if world > 1:                        # one set of initial weights: rank 0's (run_PMR_ModCR.py loads one checkpoint on every rank)
            for t in list(model.parameters()) + list(model.buffers()):
                dist.broadcast(t.data, 0)
        model.train()
        names = tu.trainable_parameters(model)
        pdict = dict(model.named_parameters())
        for k, p in pdict.items():
            p.requires_grad_(k in names)
        flat = tu.FlatGrads([pdict[k] for k in names], dev, names=names, comm_dtype=torch.bfloat16 if args.grad_comm == "bf16" else None)
        opt = tu.FlatAdamW(flat, names, t_total=100000, form=args.optimizer)
        return model, flat, opt

    model, flat, opt = setup(args.train_encoders, args.with_roberta)
    mh.DROPOUT.manual_seed(1000 + rank)

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
        main_s = torch.cuda.current_stream()
        if fetch.nxt is None:
            with torch.cuda.stream(copy_stream):
                fetch.nxt = tu.batch_to_device(host_batches[i % nb], dev)
        main_s.wait_stream(copy_stream)
        cur = fetch.nxt
        for v in cur.values():
            for t in (v if isinstance(v, list) else [v]):
                if torch.is_tensor(t):
                    t.record_stream(main_s)
        with torch.cuda.stream(copy_stream):
            fetch.nxt = tu.batch_to_device(host_batches[(i + 1) % nb], dev)
        return cur
    fetch.nxt = None

    def is_roofline_attention(x, *a, **k):
        # the dominant kernel: qkv_attn4_kernel<1> = production call with the broadcast key mask (global_enc and the
        # phase-2 layers of seq_enc, 18 of the 24 S=180 launches per step); the dense-mask / chunk-mean / align-map
        # variants are other kernels (qkv_attn4_kernel<2>, <3>) with their own rows in the rocprof summary
        return (x.shape[1] == s_len and x.shape[2] == H_OSCAR and k.get("mask_bits") is None and k.get("chunk_id") is None
                and k.get("align_map") is None and k.get("hist") is None and not k.get("want_probs"))

    def run_timed(model, flat, opt, steps, warmup, timer=None, fetch=fetch):
        for i in range(warmup):
            tu.train_step(model, fetch(i), opt, None, flat, world)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        if timer is not None:
            timer.enabled = True
        t0 = time.perf_counter()
        for i in range(steps):
            loss, logits = tu.train_step(model, fetch(warmup + i), opt, None, flat, world)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        if timer is not None:
            timer.enabled = False
        if world > 1:
            # every rank's own elapsed time (a straggler shows as max >> min); the contract's figure is the MAX over ranks
            allr = [None] * world
            dist.all_gather_object(allr, float(elapsed))
            run_timed.per_rank_s = [float(t) for t in allr]
            elapsed = max(run_timed.per_rank_s)
        return elapsed, loss
    run_timed.per_rank_s = None

    with KernelTimer(mh, "qkv_attn", is_roofline_attention) as kt:
        elapsed, loss = run_timed(model, flat, opt, args.steps, args.warmup, kt)
        t_attn = kt.mean_seconds()
    flat_buckets, launched_in_bwd = [list(b) for b in flat.buckets], flat.launched_in_backward
    per_rank_s = run_timed.per_rank_s
    if world > 1:
        # what the communicator itself saw (not the environment): an all-reduce of ones over the RCCL group, its version, and one
        # extra untimed step with every bucket's launch -> completion stamped on the compute stream
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        ranks_seen = int(round(float(ones.item())))
        try:
            rccl_version = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception as e:          # noqa: BLE001
            rccl_version = "unavailable (%s)" % type(e).__name__
        flat.timing = True
        tu.train_step(model, fetch(0), opt, None, flat, world)
        flat.timing = False
        bucket_ms = list(flat.bucket_ms)

    def attn_flops(n):
        return n * (6.0 * s_len * H_OSCAR ** 2 + 4.0 * s_len * s_len * H_OSCAR)     # SURVEY 8(d), padding not counted

    def kernel_seconds(n, train_mode, launches=20):
        """the roofline kernel by itself: `launches` back-to-back C-ABI calls after the timed region, layer-0 weights of
        global_enc, n sequences of the step's shape and padding mask; train_mode = with the attention-dropout masking"""
        att = model.calec.global_enc.encoder.layer[0].attention.self
        w, b = att.packed_qkv(torch.bfloat16)
        x = torch.randn(n, s_len, H_OSCAR, device=dev).to(torch.bfloat16)
        km = batches[0]["input_mask"].to(torch.float32)
        km = km[:n] if km.shape[0] >= n else km.repeat((n + km.shape[0] - 1) // km.shape[0], 1)[:n]
        drop = (args.attn_dropout, 17, 4242) if train_mode else None
        for _ in range(3):
            mh.qkv_attn(x, w, b, key_mask=km, num_heads=A_OSCAR, attn_dropout=drop)
        kt2 = KernelTimer(mh, "qkv_attn", lambda *a, **k: True)
        with kt2:
            kt2.enabled = True
            for _ in range(launches):
                mh.qkv_attn(x, w, b, key_mask=km, num_heads=A_OSCAR, attn_dropout=drop)
        torch.cuda.synchronize()
        return kt2.mean_seconds()

    def prefix_call_seconds(n=256, pfx=10, launches=20):
        """the north-star call shape with prefix rows: K, V over [prefix(10) ; x(170)] = 180 keys, queries from x
        (modeling_bert.py:36-44): whole C-ABI call (row concatenation into the workspace + tile kernel), back to back"""
        att = model.calec.global_enc.encoder.layer[0].attention.self
        w, b = att.packed_qkv(torch.bfloat16)
        sq = s_len - pfx
        x = torch.randn(n, sq, H_OSCAR, device=dev).to(torch.bfloat16)
        hist = torch.randn(n, pfx, H_OSCAR, device=dev).to(torch.bfloat16)
        km = torch.ones(n, s_len, device=dev)
        for _ in range(3):
            mh.qkv_attn(x, w, b, key_mask=km, hist=hist, num_heads=A_OSCAR)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(launches):
            mh.qkv_attn(x, w, b, key_mask=km, hist=hist, num_heads=A_OSCAR)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / launches * 1e-3
        fl = n * (2.0 * sq * H_OSCAR ** 2 + 4.0 * s_len * H_OSCAR ** 2 + 4.0 * sq * s_len * H_OSCAR)
        return t, fl, sq, pfx

    def pmc_traffic(pmc_file):
        """HBM-side bytes per launch of the roofline kernel FROM THE COMMITTED rocprofv3 --pmc passes (profiles/<pmc_file>:
        FETCH_SIZE, WRITE_SIZE in KB; FETCH_SIZE doubled as the gfx950 guide prescribes; N=256, S=180, H=768).  Not live:
        counters need their own rocprofv3 runs."""
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

    out = None
    if rank == 0:
        attn_drop = args.attn_dropout > 0
        achieved = attn_flops(n_seq) / t_attn / 1e12 if t_attn else None
        pmc_file = None
        # (round 5: the counter passes of the FINAL kernels, tools/run_pmc_r05.sh; older rounds' files only as a fallback)
        for cand in (("r05_pmc_attn_n256_train.txt", "r02_attn4_drop_pmc.txt", "r01_attn4_drop_pmc.txt") if attn_drop
                     else ("r05_pmc_attn_n256_eval.txt", "r02_attn4_pmc.txt", "r01_attn4_pmc.txt")):
            if os.path.exists(os.path.join(ROOT, "profiles", cand)):
                pmc_file = cand
                break
        traffic256 = pmc_traffic(pmc_file) if pmc_file else None
        workload = ("%s 4-choice T=%d R=%d (S=%d) H=%d L=%d, %d examples (=%d sequences)/GPU/step: %s, cls_layer_lyx x2 + mapping networks + scorer + "
                    "MC-CE fwd+bwd, grad clip + AdamW (transformers.AdamW arithmetic, fused flat-buffer step); %s; %s" % (
                        {"pmr": "PMR", "c5": "VCR-like (BASELINE configs[4] shape class, Oscar-large)", "toy": "TOY dims (plumbing rehearsal, not a measurement)"}[args.config], T_TEXT, R_IMG, s_len, H_OSCAR, L_OSCAR, args.batch, n_seq,
                        "Oscar-base global_enc (full S=180) + seq_enc fwd+BWD with gradients (--train-encoders, SURVEY 8f-4), image-only global_enc pass S=101 fwd"
                        if args.train_encoders else "frozen global_enc (image-only S=%d + full S=%d) + seq_enc fwd" % (1 + R_IMG, s_len),
                        "prefix RoBERTa-large body INCLUDED (24 layers, H=1024, S=106, fwd+bwd, trainable)" if args.with_roberta
                        else "prefix-RoBERTa body NOT included (stand-in pooler; SURVEY 8f-1 'next', --with-roberta adds it)",
                        ("hidden dropout %.2g live (embeddings, BertSelfOutput, BertOutput, heads, cross_attention_lyx weights 0.1; counter-based masks), "
                         "attention-probability dropout %.2g live" % (args.dropout, args.attn_dropout)) if args.dropout > 0 else "dropout off"))
        out = {
            "metric": {"pmr": "PMR training examples/sec (4-choice, seq~180)", "c5": "VCR-like training examples/sec (4-choice, seq=230, H=1024)",
                       "toy": "toy-dims examples/sec (plumbing rehearsal, not a measurement)"}[args.config],
            "value": round(args.batch * world * args.steps / elapsed, 3),
            "unit": "examples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": workload, "global_batch": args.batch * world, "seq_len": s_len, "parallelism": "dp%d" % world,
                       "optimizer": args.optimizer,
                       "inputs": {"none": "resident in HBM before the timed region", "sync": "PCIe-INCLUSIVE: copied from pinned host memory before every step",
                                  "overlap": "PCIe-INCLUSIVE: copied from pinned host memory on a side stream under the previous step"}[args.h2d]},
            "loss": round(float(loss.item()), 5),
        }
        if args.last_layer_rows:
            out["config"]["workload"] += ("; OPT-IN --last-layer-rows: the frozen encoders' last layers skip the token-wise blocks of the rows ModCR "
                                          "never reads (same loss / logits / gradients; not the default)")
            out["config"]["last_layer_rows"] = True
        if knobs:
            out["config"]["env_knobs"] = {k: os.environ[k] for k in knobs}
        if per_rank_s:
            ms = [t / args.steps * 1e3 for t in per_rank_s]
            out["ms_per_step_by_rank"] = {"min": round(min(ms), 3), "max": round(max(ms), 3), "ranks": [round(v, 3) for v in ms],
                                          "host_threads_per_rank": torch.get_num_threads()}
        if args.rehearse_on_one_gpu:
            out["config"]["rehearsal"] = "all %d ranks share ONE GPU, collectives over gloo: not a measurement" % world
        if achieved:
            # (the five template parameters as rocprofv3 prints them: KMODE, token tile, DROP, heads per workgroup, DUMPV)
            kname = ("qkv_attn4_kernel<1, 192, %d, 2, 0>" % (1 if attn_drop else 0)) if 128 < s_len <= 192 else \
                    ("qkv_attn4_kernel<1, 256, %d, 1, 0> (256-token tile, one head per workgroup)" % (1 if attn_drop else 0))
            out["roofline"] = {"kernel": "%s (fused QKV projection + attention fwd%s, N=%d S=%d H=%d)"
                                         % (kname, ", training mode: attention-probability dropout mask applied in the kernel"
                                            if attn_drop else "", n_seq, s_len, H_OSCAR),
                               "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16 / 1e12,
                               "unit": "TFLOP/s", "frac": round(achieved * 1e12 / PEAK_BF16, 4),
                               "launches_timed": len(kt.pairs), "avg_launch_us": round(t_attn * 1e6, 2),
                               "where": "HIP events stamped by hipExtLaunchKernel at the start / end of every such launch inside the timed region",
                               "algorithmic_gflop_per_launch": round(attn_flops(n_seq) / 1e9, 2),
                               "traffic": (round(traffic256 * n_seq / 256.0) if traffic256 else None),
                               "traffic_source": ("NOT measured in this run: 2*FETCH_SIZE + WRITE_SIZE of the committed rocprofv3 --pmc pass profiles/%s "
                                                  "(N=256), scaled by N/256" % pmc_file) if traffic256 else None}
            # the in-step size has a counter pass of its own (round 5: tools/run_pmc_r05.sh, N = 512, training-mode variant, final kernel)
            pmc512 = "r05_pmc_attn_n512.txt" if os.path.exists(os.path.join(ROOT, "profiles", "r05_pmc_attn_n512.txt")) else "r04_pmc_attn_n512.txt"
            if attn_drop and n_seq == 512 and os.path.exists(os.path.join(ROOT, "profiles", pmc512)) and pmc_traffic(pmc512):
                out["roofline"]["traffic"] = pmc_traffic(pmc512)
                out["roofline"]["traffic_source"] = ("NOT measured in this run: 2*FETCH_SIZE + WRITE_SIZE of the committed rocprofv3 --pmc pass "
                                                     "profiles/%s (same kernel and size, N=512)" % pmc512)
            # BASELINE configs[1]: N = 256 sequences, fused attention fwd only
            if args.config != "pmr":
                out["roofline"]["traffic"] = out["roofline"]["traffic_source"] = None
            c2 = {"shape": "N=256 S=%d H=%d A=%d (BASELINE configs[1]: batch=64, fused prefix-attention fwd only)" % (s_len, H_OSCAR, A_OSCAR),
                  "algorithmic_gflop_per_launch": round(attn_flops(256) / 1e9, 2), "where": "20 back-to-back launches after the timed region",
                  "traffic": traffic256, "traffic_source": "profiles/%s (committed PMC pass, not live)" % pmc_file if traffic256 else None}
            for nm, tm in (("eval", False), ("train", True)):
                if (tm and not attn_drop) or args.config != "pmr":
                    continue
                t = kernel_seconds(256, tm)
                c2[nm] = {"kernel": "qkv_attn4_kernel<1, 192, %d, 2, 0>" % (1 if tm else 0), "avg_launch_us": round(t * 1e6, 2),
                          "achieved": round(attn_flops(256) / t / 1e12, 2), "frac": round(attn_flops(256) / t / PEAK_BF16, 4)}
            if args.config == "pmr":
                out["roofline"]["config2"] = c2
                tp, flp, sq, pfx = prefix_call_seconds()
                out["roofline"]["prefix_rows"] = {
                    "shape": "N=256, prefix P=%d + S=%d query rows (keys %d), H=%d A=%d, eval mode" % (pfx, sq, s_len, H_OSCAR, A_OSCAR),
                    "avg_call_us": round(tp * 1e6, 2), "algorithmic_gflop_per_call": round(flp / 1e9, 2),
                    "achieved": round(flp / tp / 1e12, 2), "frac": round(flp / tp / PEAK_BF16, 4),
                    "where": "20 back-to-back modcr_qkv_attn_fwd calls with history_state after the timed region; the call = one row-concatenation "
                             "launch into the workspace + qkv_attn4_kernel<1,192,0> over the 180-row tile (torch events around the loop)"}

    # ---- secondary workloads of the same line, each with its own model / optimizer / batches, ms per step and the in-step fused
    # attention fraction of ITS main attention shape (kernel-exact events, as `roofline`):
    #   config3_full_fwd_bwd  BASELINE configs[2]: the same step with both Oscar encoders trained (every layer's backward on the HIP kernels)
    #   with_roberta          the reference's REAL step: + the 24-layer prefix RoBERTa-large body forward and backward
    #   c5                    BASELINE configs[4]'s shape class: Oscar-large H=1024, 16 heads, 24 layers, S=230 (VCR), 32 examples
    def leg(train_encoders, with_roberta, cfg_name, min_steps, workload, last_layer_rows=None):
        c = CONFIGS[cfg_name]
        t_, r_, h_, a_, l_ = c["T"], c["R"], c["H"], c["A"], c["L"]
        bsz = args.batch if cfg_name == args.config else c["batch"]
        t_build = time.perf_counter()
        m2, f2, o2 = setup(train_encoders, with_roberta, dims=(h_, l_, a_), last_layer_rows=last_layer_rows)
        b2 = [tu.batch_to_device(synthetic.make_batch(bsz, T=t_, R=r_, seed=4321 + 97 * rank + i), dev) for i in range(2)]
        fetch2 = lambda i: b2[i % len(b2)]
        s2 = t_ + r_

        rob_shape = {}

        def sel(x, *a, **k):
            if with_roberta and x.shape[2] == 1024 and x.shape[1] <= 128 and k.get("mask_bits") is None:
                rob_shape.update(S=int(x.shape[1]), H=int(x.shape[2]))
            if with_roberta and x.shape[2] == 1024 and x.shape[1] <= 128 and k.get("mask_bits") is None:
                return "roberta"                    # the prefix RoBERTa-large body's layers (S = 96 + 10, H = 1024, 16 heads): lse + Q|K|V dump variant
            return (x.shape[1] == s2 and x.shape[2] == h_ and k.get("mask_bits") is None and k.get("chunk_id") is None
                    and k.get("align_map") is None and k.get("hist") is None and not k.get("want_probs"))
        # one untimed step sizes the run: about --leg-seconds, at least min_steps
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tu.train_step(m2, fetch2(0), o2, None, f2, world)
        torch.cuda.synchronize()
        tu.train_step(m2, fetch2(1), o2, None, f2, world)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / 2.0
        st = int(max(min_steps, min(50, args.leg_seconds / max(per, 1e-3))))
        if world > 1:                            # every rank must run the same number of steps
            stt = torch.tensor([st], device=dev)
            dist.broadcast(stt, 0)
            st = int(stt.item())
        with KernelTimer(mh, "qkv_attn", sel) as kt2:
            el, ls = run_timed(m2, f2, o2, st, 1, kt2, fetch=fetch2)
            ta = kt2.mean_seconds()
        n2 = bsz * 4
        fl = n2 * (6.0 * s2 * h_ ** 2 + 4.0 * s2 * s2 * h_)
        res = {"ms_per_step": round(el / st * 1e3, 3), "value": round(bsz * world * st / el, 3), "unit": "examples/s", "steps": st,
               "warmup": 3, "loss": round(float(ls.item()), 5), "examples_per_gpu": bsz, "workload": workload,
               "build_seconds": round(t0 - t_build, 1)}
        if ta:
            res["in_step_attention"] = {"shape": "N=%d S=%d H=%d A=%d, key-mask variant, training mode" % (n2, s2, h_, a_),
                                        "launches_timed": len(kt2.pairs), "avg_launch_us": round(ta * 1e6, 2),
                                        "achieved": round(fl / ta / 1e12, 2), "frac": round(fl / ta / PEAK_BF16, 4)}
        tr = kt2.mean_seconds("roberta")
        if tr:
            sr, hr = rob_shape["S"], rob_shape["H"]
            flr = n2 * (6.0 * sr * hr ** 2 + 4.0 * sr * sr * hr)
            res["in_step_attention_roberta"] = {
                "shape": "N=%d S=%d H=%d A=%d: the trainable prefix RoBERTa-large body's fused attention forward (72 %% of this step's FLOPs sit in that "
                         "body), 128-token tile, key mask, attention dropout 0.1, row statistics + Q|K|V image dump for the backward" % (n2, sr, hr, hr // 64),
                "kernel": "qkv_attn4_kernel<1, 128, 1, 2, 1>", "launches_timed": len(kt2.labelled["roberta"]), "avg_launch_us": round(tr * 1e6, 2),
                "algorithmic_gflop_per_launch": round(flr / 1e9, 2), "achieved": round(flr / tr / 1e12, 2), "frac": round(flr / tr / PEAK_BF16, 4)}
        if with_roberta and rank == 0 and world == 1 and not args.no_cpu_baseline and cfg_name == "pmr":
            # the like-for-like figure for run_PMR_ModCR.py:201-227 gets its own CPU baseline and agreement check
            res["cpu_baseline"], res["parity_vs_oracle"] = real_step_evidence(m2, dev, max(1, min(32, usable_cpus()[1])), args.real_step_seconds)
        del m2, f2, o2, b2
        torch.cuda.empty_cache()
        return res

    plain = not args.train_encoders and not args.with_roberta and args.h2d == "none" and args.config == "pmr"
    if plain and not args.no_config3:
        del opt, flat
        r3 = leg(True, False, "pmr", 10,
                 "the same step with global_enc (full pass) and seq_enc TRAINED: 24 encoder layers forward + backward on the HIP kernels "
                 "(five-product attention backward on the forward's row statistics, LayerNorm / GELU backward), image-only pass forward, "
                 "heads, clip + AdamW over all parameters")
        if rank == 0:
            out["config3_full_fwd_bwd"] = r3
    if plain and not args.no_extra_legs and world == 1:
        out["with_roberta"] = leg(False, True, "pmr", 5,
                                  "the reference's real training step (SURVEY 8f-1): frozen Oscar encoders + heads as the headline, PLUS the 24-layer "
                                  "prefix RoBERTa-large body (H=1024, 16 heads, S=106) forward and backward on the same kernels, trainable, its own "
                                  "dropouts 0.1 / 0.1 live; random-init weights (the checkpoint is not in the reference tree)")
        if not args.last_layer_rows:
            out["last_layer_rows"] = leg(False, False, "pmr", 10,
                                         "the HEADLINE step with config.modcr_last_layer_rows (opt-in, off in `value`): the last layer of each frozen encoder "
                                         "pass runs BertSelfOutput / BertIntermediate / BertOutput only over the rows ModCR reads (text rows of the two full "
                                         "passes, the [CLS] row of the image-only pass; attention still over all rows) -- loss, logits and gradients are those "
                                         "of the full computation (tests/test_hip_models.py::test_last_layer_rows_opt_in_changes_no_consumed_value)",
                                         last_layer_rows=True)
        out["c5"] = leg(False, False, "c5", 10,
                        "BASELINE configs[4] shape class (run_vcr_ModCR.py, Oscar-large): T=194 + R=36 = S 230, H=1024, 16 heads, 24 layers, frozen "
                        "encoders + heads, 32 examples = the reference's 8 x 4 accumulation; the 256-token tile kernels")
    if world > 1 and rank == 0:
        out["rccl_ranks_seen"] = ranks_seen
        out["config"]["collective"] = {"backend": "gloo (one-GPU rehearsal)" if args.rehearse_on_one_gpu else "nccl (= RCCL on ROCm) %s" % rccl_version,
                                       "ranks_seen_by_all_reduce": ranks_seen,
                                       "bucket_launch_to_complete_ms": bucket_ms,
                                       "bucket_timing": "one extra untimed step: event at each bucket's launch (post-accumulate-grad hook, compute stream) "
                                                        "and behind the stream-side wait for its all-reduce in finish(); later buckets include the wait "
                                                        "for earlier ones",
                                       "expected_all_reduce_ms": "DESIGN.md section 6: 242 MB of fp32 gradients here (1.66 GB with the RoBERTa body) over 7 xGMI links"}
        out["config"]["gradient_buckets"] = {"bytes": [int((e - s_) * 4) for s_, e, _ in flat_buckets], "count": len(flat_buckets),
                                             "launched_during_backward_last_step": launched_in_bwd, "comm_dtype": args.grad_comm,
                                             "note": "flat fp32 gradient buffer in reverse registration order, one asynchronous RCCL all-reduce per "
                                                     "bucket launched from post-accumulate-grad hooks while backward is still running"}

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline and not args.with_roberta and not args.train_encoders and args.config == "pmr":
            mask_cores, ncpu = usable_cpus()
            nthreads = max(1, min(32, ncpu))
            out["cpu_baseline"] = cpu_baseline(model, 4321, nthreads, host_cores=ncpu, affinity_cores=mask_cores)
            out["parity_vs_oracle"] = agreement_rate(model, dev, args.parity_examples, args.parity_seconds, nthreads)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
