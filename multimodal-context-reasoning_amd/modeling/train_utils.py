"""Model assembly and one optimisation step, shared by run_PMR_ModCR.py, run_vcr_ModCR.py and
bench.py.  Mirrors the reference's main()/train() (run_PMR_ModCR.py:709-802, :127-145, :201-227):
two frozen Oscar-base encoders inside ChunkAlign_CLS_enc4_align_ensemble, Abstract_Specific on top,
AdamW with two learning-rate groups ('seq_enc' x0.1), eps 1e-5, linear decay without warm-up,
clip_grad_norm_(all, 1.0) every micro-step.

Multi-GPU = pure data parallel over examples: one process per GPU, the trainable
parameters' gradients live in ONE flat fp32 buffer that is all-reduced (RCCL over xGMI) in a few large
buckets launched from gradient hooks while backward is still running; the frozen encoders never enter
the collective.
"""
import os
import weakref

import torch
import torch.distributed as dist

from .bert_primitives import BertConfig
from .modeling_ensemble import Abstract_Specific
from .modeling_transfomres import BertImgModel
from .modeling_vcr_chunkalign_v10 import ChunkAlign_CLS_enc4_align_ensemble, SeqBertImgModel
from .roberta_prefix import PrefixPoolerStandIn


def usable_cpus():
    """(cores in this process's affinity mask, CPUs the cgroup lets it use): the GPU boxes of this pool show 256 cores in the mask under
    a 16-CPU quota (cpu.max = 1600000 100000) -- threads beyond the quota only wait for each other"""
    try:
        ncpu = len(os.sched_getaffinity(0))
    except AttributeError:
        ncpu = os.cpu_count() or 1
    quota = ncpu
    try:
        txt = open("/sys/fs/cgroup/cpu.max").read().split()
        if txt and txt[0] != "max":
            quota = max(1, int(float(txt[0]) / float(txt[1]) + 0.5))
    except (OSError, ValueError, IndexError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0 and per > 0:
                quota = max(1, int(q / per + 0.5))
        except (OSError, ValueError):
            pass
    return ncpu, min(ncpu, quota)


def cap_host_threads(local_world):
    """one process per GPU: the ranks of a node share its CPU quota.  torch's default intra-op pool is one thread per core in the
    affinity mask (256 on this pool's GPU boxes, under a 16-CPU cgroup quota), i.e. N x 256 threads on 16 CPUs for every host-side
    torch op of the launch loop; each rank takes its share of the quota instead (run_PMR_ModCR.py:423-448 leaves this to the launcher's
    OMP_NUM_THREADS, which torch.distributed.run sets to 1 -- an explicit OMP_NUM_THREADS is respected here too)."""
    if os.environ.get("OMP_NUM_THREADS"):
        return torch.get_num_threads()
    n = max(1, usable_cpus()[1] // max(1, int(local_world)))
    torch.set_num_threads(n)
    return n


def oscar_config(vocab_size=30522 + 45, dtype="bf16", hidden_size=768, num_attention_heads=12, attention_probs_dropout_prob=0.0,
                 num_hidden_layers=12, hidden_dropout_prob=0.0, max_hypo=50, add_residual=False, add_local_residual=False, **kw):
    """config edits of run_PMR_ModCR.py:717-726 / :736-748.  hidden_dropout_prob = args.drop_out there (0.3): applied in
    training mode by the embeddings, BertSelfOutput, BertOutput and the trainable heads (the
    attention-probability dropout is not)."""
    d = dict(vocab_size=vocab_size, hidden_size=hidden_size, num_attention_heads=num_attention_heads,
             num_hidden_layers=num_hidden_layers, intermediate_size=4 * hidden_size,
             img_feature_dim=2054, img_feature_type="frcnn", use_img_layernorm=1, img_layer_norm_eps=1e-12,
             hidden_dropout_prob=hidden_dropout_prob, attention_probs_dropout_prob=attention_probs_dropout_prob, output_attentions=True,
             output_hidden_states=False, max_hypo=max_hypo, add_residual=add_residual, add_local_residual=add_local_residual,
             modcr_dtype=dtype)
    d.update(kw)                        # (tests: img_feature_dim, max_position_embeddings of a small twin)
    return BertConfig(**d)


def build_model(device, vocab_size=30522 + 45, dtype="bf16", seed=0, roberta_model=None, roberta_body="standin",
                hidden_dropout_prob=0.0, train_encoders=False, attention_probs_dropout_prob=0.0, roberta_hidden_dropout_prob=0.0,
                hidden_size=768, num_hidden_layers=12, num_attention_heads=None, **cfg_kw):
    """roberta_body: "standin" (small trainable pooler over the prefix, the default of bench.py) or "large" = the
    24-layer prefix RoBERTa-large on the HIP kernels, trainable end to end as in run_PMR_ModCR.py:772-781."""
    torch.manual_seed(seed)
    if roberta_model is None and roberta_body == "large":
        from .roberta_prefix import RobertaPrefixModel
        roberta_model = RobertaPrefixModel(attention_probs_dropout_prob=attention_probs_dropout_prob, hidden_dropout_prob=roberta_hidden_dropout_prob)
    kw = dict(hidden_dropout_prob=hidden_dropout_prob, attention_probs_dropout_prob=attention_probs_dropout_prob, hidden_size=hidden_size,
              num_hidden_layers=num_hidden_layers, num_attention_heads=num_attention_heads or hidden_size // 64)
    kw.update(cfg_kw)                   # max_hypo, add_residual, add_local_residual (run_PMR_ModCR.py:743-745)
    cfg_g = oscar_config(vocab_size, dtype, **kw)
    cfg_s = oscar_config(vocab_size, dtype, **kw)
    oscar_model = BertImgModel(cfg_g)
    seq_model = SeqBertImgModel(cfg_s)
    calec = ChunkAlign_CLS_enc4_align_ensemble(oscar_model, seq_model, num_labels=4)
    calec.set_train_encoders(train_encoders)
    if roberta_model is None:
        roberta_model = PrefixPoolerStandIn()
    model = Abstract_Specific(calec_model=calec, clip_model=None, roberta_model=roberta_model, num_labels=4)
    return model.to(device)


def trainable_parameters(model):
    """Parameters that receive a gradient in the reference's step: everything
    outside the two no_grad encoders that the forward actually uses."""
    names = []
    train_enc = getattr(getattr(model, "calec", None), "train_encoders", False)
    for k, p in model.named_parameters():
        if k.startswith("calec.global_enc.") or k.startswith("calec.seq_enc."):
            if train_enc and not k.endswith("edge_dense.weight"):       # set_train_encoders(): SURVEY 8f-4
                names.append(k)
            continue
        used = (k.startswith("roberta.") or k.startswith("mapping_network_") or k.startswith("abst_confidence_scorer")
                or k.startswith("calec.cls_ensemble_1")
                or (k.startswith("calec.cls_layer_lyx.") and (".cross_attention." in k or ".LayerNorm." in k
                                                              or ".intermediate." in k or ".output." in k)
                    and ".attention." not in k))
        if used and p.requires_grad:            # (run_vcr_ModCR.py:781-787 freezes the RoBERTa body by requires_grad)
            names.append(k)
    return names


class FlatGrads(object):
    """One contiguous fp32 gradient buffer; p.grad are views into it (no copies before the collective).

    Multi-GPU: the buffer is laid out in REVERSE registration order (= roughly the order backward produces the
    gradients: mapping networks and scorer first, cls_ensemble_1 last) and cut into buckets of about
    `bucket_bytes`.  A post-accumulate-grad hook per parameter counts a bucket down; when its last gradient
    has landed the bucket's slice is all-reduced asynchronously (RCCL runs it on its own stream, over xGMI)
    while autograd is still producing the later buckets.  finish() launches whatever backward did not reach,
    waits for every bucket and scales to the global-batch mean.  all_reduce() is the non-overlapped single
    collective (kept for A/B and for callers that fill the gradients by hand)."""

    ALIGN = 64          # elements

    QKV = ("attention.self.query.", "attention.self.key.", "attention.self.value.")

    def __init__(self, params, device, bucket_bytes=64 << 20, names=None, comm_dtype=None):
        # comm_dtype (opt-in, torch.bfloat16): every bucket crosses the links as a bf16 copy -- half the bytes of the 1.66 GB the real
        # step (RoBERTa body trainable) all-reduces, for 8 significant bits per summand (the sum itself is taken by the collective in
        # that type); the flat buffer, the clip and AdamW stay fp32 and every rank still ends with identical values
        self.comm_dtype = comm_dtype
        self._comm = {}
        self.params = list(params)
        order = list(reversed(self.params))
        if names is not None:
            # the q | k | v weight gradients of an attention block come out of ONE [3H, H] product (and the biases out of one [3H] column
            # sum): laid out query, key, value next to one another they are one span of the buffer the kernel can write (take_span);
            # the six tensors keep their place in the reverse-order walk as a group
            names = list(reversed(list(names)))
            tail = "attention.self.value.bias"
            i = 0
            while i + 6 <= len(order):
                pre = names[i][:-len(tail)] if names[i].endswith(tail) else None
                # reversed registration order of one block: value.bias, value.weight, key.bias, key.weight, query.bias, query.weight
                if pre is not None and names[i:i + 6] == [pre + q + t for q in reversed(self.QKV) for t in ("bias", "weight")]:
                    byname = dict(zip(names[i:i + 6], order[i:i + 6]))
                    order[i:i + 6] = [byname[pre + q + t] for t in ("weight", "bias") for q in self.QKV]
                    i += 6
                else:
                    i += 1
        self.order = order                     # layout order (increasing offset)
        # every tensor starts on a 256-byte boundary: the kernels' 16-byte loads of biases / LayerNorm parameters and the
        # persistent GEMMs' operand checks need 16 (one [1]-shaped bias would otherwise leave everything behind it 4-byte aligned);
        # the padding stays zero in the gradients, the parameters and both moments
        self.offsets, total = {}, 0
        for p in order:
            self.offsets[id(p)] = total
            total += (p.numel() + self.ALIGN - 1) // self.ALIGN * self.ALIGN
        self.flat = torch.zeros(total, dtype=torch.float32, device=device)
        self.buckets = []                      # [start, end, number of parameters]
        self._bucket_of = {}
        off, b_start, b_n = 0, 0, 0
        for p in order:
            n = p.numel()
            off = self.offsets[id(p)]
            p.grad = self.flat[off:off + n].view_as(p)
            self._bucket_of[id(p)] = len(self.buckets)
            off += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            b_n += 1
            if (off - b_start) * 4 >= bucket_bytes:
                self.buckets.append([b_start, off, b_n])
                b_start, b_n = off, 0
        if b_n:
            self.buckets.append([b_start, off, b_n])
        self._armed = False
        self._counted = set()
        self._left, self._works, self.launched_in_backward = [], [], 0
        self._written = set()                  # parameters whose slice a backward kernel has written in place since zero()
        self._uses = {}                        # forward nodes per parameter in the current graph (note_use; cleared by finish() / zero())
        self._done_n = {}                      # ... of which have reported their contribution in this backward (done)
        self.in_place = True                   # gradient sink of hip_autograd.BertLayerFn (see there); False: every gradient through autograd
        self.install()                         # (the newest FlatGrads is the sink; a caller that switches between two re-installs)
        self.timing = False                    # bench.py (N > 1): stamp every bucket's launch and completion on the compute stream
        self.bucket_ms = []
        self._ev = []
        ref = weakref.ref(self)                # (the hooks live as long as the Parameters: they must not keep this buffer alive)

        def hook(p, _r=ref):
            o = _r()
            if o is not None:
                FlatGrads._on_grad(o, p)
        for p in self.params:
            if hasattr(p, "register_post_accumulate_grad_hook"):
                p.register_post_accumulate_grad_hook(hook)

    def install(self):
        """make this buffer the gradient sink of the HIP autograd functions (held there by weak reference)"""
        from . import hip_autograd as _ag
        _ag.set_grad_sink(self)

    def close(self):
        """stop being the sink (gradients of this buffer's parameters go through autograd again)"""
        from . import hip_autograd as _ag
        if _ag.grad_sink() is self:
            _ag.set_grad_sink(None)

    def zero(self):
        self.flat.zero_()
        self._written.clear()
        self._uses.clear()
        self._done_n.clear()

    def new_graph(self):
        """call before a training forward (micro_step does): use counts left by an earlier graph whose backward never ran -- an eval
        forward without no_grad, an exception between forward and backward, a tools loop that calls backward without finish / zero --
        belong to that dead graph and are dropped, so that this forward's nodes are counted afresh (a stale count of 2 would refuse
        every single_use take and keep done() from ever reaching the use count: correct gradients through autograd, but the in-place
        sink and the all-reduce overlap silently lost)"""
        self._uses.clear()
        self._done_n.clear()

    def note_use(self, p):
        """a forward node that will produce a gradient for p was created (hip_autograd._note_uses)"""
        self._uses[id(p)] = self._uses.get(id(p), 0) + 1

    # ---- gradient sink (hip_autograd.GRAD_SINK) --------------------------------------------------------------
    def take(self, p, accumulates=False, single_use=False):
        """the slice of the flat buffer a backward kernel may use for p's gradient, or None.  A kernel that ACCUMULATES (the LayerNorm
        parameter gradients) can always use it; one that WRITES only while nothing has been added to the slice since zero().
        single_use: only for a parameter that exactly one forward node has used since zero() (the caller will report it done)."""
        if not self.in_place or id(p) not in self.offsets or p.grad is None:
            return None
        if single_use and self._uses.get(id(p), 0) != 1:
            return None
        off = self.offsets[id(p)]
        if p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
            return None                                # .grad was re-assigned by somebody: leave it to autograd
        if not accumulates:
            if id(p) in self._written:
                return None                            # second micro-batch of an accumulation window: autograd adds
        self._written.add(id(p))
        return p.grad

    def untake(self, p):
        self._written.discard(id(p))

    def take_span(self, ps, single_use=False):
        """one WRITABLE tensor covering the gradients of `ps` (laid out back to back, in this order, without padding: the q | k | v
        weights or biases of one attention block when the buffer was built with `names`), or None"""
        if not self.in_place or any(id(p) not in self.offsets or p.grad is None or id(p) in self._written for p in ps):
            return None
        if single_use and any(self._uses.get(id(p), 0) != 1 for p in ps):
            return None
        off0 = off = self.offsets[id(ps[0])]
        for p in ps:
            if self.offsets[id(p)] != off or p.grad.data_ptr() != self.flat.data_ptr() + 4 * off:
                return None
            off += p.numel()
        for p in ps:
            self._written.add(id(p))
        return self.flat[off0:off]

    def done(self, p):
        """a backward node has put its contribution for p into the buffer.  The bucket is counted down when the LAST forward node of
        p in this graph has reported; a parameter with two nodes of which one goes through autograd is counted by autograd's hook,
        which fires once, after both have run."""
        n = self._done_n.get(id(p), 0) + 1
        self._done_n[id(p)] = n
        if n >= self._uses.get(id(p), 1):
            self._on_grad(p)

    def all_reduce(self, world_size):
        if world_size > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(world_size)

    # ---- overlapped form ---------------------------------------------------------------------------------
    def begin(self, world_size, force=False):
        """call before loss.backward().  force: run the bucketed collectives even on one rank (tools/nccl_world1_check.py: the
        hook-launched RCCL all-reduces under the HIP backward without a second GPU)"""
        self._armed = world_size > 1 or force
        self._left = [b[2] for b in self.buckets]
        self._works = [None] * len(self.buckets)
        self.launched_in_backward = 0
        self._ev = [None] * len(self.buckets)
        self._counted = set()                  # a parameter counts a bucket down once per backward (sink report or autograd hook)
        self._done_n = {}

    def _launch(self, b):
        s, e, _ = self.buckets[b]
        if self.timing and self.flat.is_cuda:
            self._ev[b] = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
            self._ev[b][0].record()
        if self.comm_dtype is not None:
            self._comm[b] = self.flat[s:e].to(self.comm_dtype)
            self._works[b] = dist.all_reduce(self._comm[b], op=dist.ReduceOp.SUM, async_op=True)
            return
        self._works[b] = dist.all_reduce(self.flat[s:e], op=dist.ReduceOp.SUM, async_op=True)

    def _on_grad(self, p):
        if not self._armed or id(p) in self._counted:
            return
        self._counted.add(id(p))
        b = self._bucket_of[id(p)]
        self._left[b] -= 1
        if self._left[b] == 0 and self._works[b] is None:
            self._launch(b)
            self.launched_in_backward += 1

    def finish(self, world_size):
        """call after loss.backward(): every bucket reduced, gradients = global-batch mean"""
        self._uses.clear()                     # the graph is consumed: the next forward's nodes are counted afresh
        self._done_n.clear()
        if world_size <= 1 and not self._armed:
            return
        if not self._armed:                    # begin() was not called: plain collective
            return self.all_reduce(world_size)
        for b in range(len(self.buckets)):
            if self._works[b] is None:         # a parameter of this bucket received no gradient in this step
                self._launch(b)
        for b, w in enumerate(self._works):
            w.wait()
            if b in self._comm:                # reduced-precision bucket: back into the fp32 buffer
                s_, e_, _ = self.buckets[b]
                self.flat[s_:e_].copy_(self._comm.pop(b))
            if self.timing and self._ev[b] is not None:
                self._ev[b][1].record()        # behind the stream-side wait for this bucket's collective
        self._armed = False
        self.flat.div_(world_size)
        if self.timing and self.flat.is_cuda:
            torch.cuda.synchronize()
            self.bucket_ms = [round(ev[0].elapsed_time(ev[1]), 3) if ev is not None else None for ev in self._ev]


def lr_lambda(scheduler, warmup_steps, t_total):
    """transformers.get_linear_schedule_with_warmup / get_constant_schedule_with_warmup (run_PMR_ModCR.py:138-145):
    the multiplier of the base learning rate after `step` scheduler steps."""
    warmup_steps = int(warmup_steps)
    if scheduler == "constant":
        return lambda step: float(step) / float(max(1.0, warmup_steps)) if step < warmup_steps else 1.0
    if scheduler == "linear":
        return lambda step: (float(step) / float(max(1, warmup_steps)) if step < warmup_steps else
                             max(0.0, float(t_total - step) / float(max(1, t_total - warmup_steps))))
    raise ValueError("Unknown scheduler type: {}".format(scheduler))


def reference_param_order(model):
    """Parameter indices of the reference's optimizer (run_PMR_ModCR.py:127-137): group 0 = named_parameters() without
    'seq_enc' in the name, group 1 = those with it; torch numbers them consecutively in that order.  Every parameter
    has requires_grad=True there (the encoders are frozen by no_grad only), so all of them are listed."""
    names = [n for n, _ in model.named_parameters()]
    g0 = [n for n in names if "seq_enc" not in n]
    g1 = [n for n in names if "seq_enc" in n]
    return g0, g1


class FlatAdamW(object):
    """clip_grad_norm_(all, max_norm) + AdamW step (run_PMR_ModCR.py:216,224-227) as two HIP kernels
    over flat buffers: the parameters are re-homed as views of one flat fp32 buffer laid out like FlatGrads'
    gradients, exp_avg / exp_avg_sq are flat too.  Arithmetic (`form`): "hf" = transformers.AdamW with
    correct_bias=True, the optimizer the reference constructs (run_PMR_ModCR.py:24,137: denom = sqrt(v) + eps,
    step = lr sqrt(bc2) / bc1) -- the default; "torch" = torch.optim.AdamW (A/B only).  Clipping =
    torch.nn.utils.clip_grad_norm_ (the global norm never leaves the device); the learning rate follows the
    reference's schedule (linear / constant with warm-up, lr_lambda).  Parameters whose name contains 'seq_enc' run
    at lr * 0.1 (run_PMR_ModCR.py:127-136) -- one launch per run of equal learning rate.

    state_dict() carries the flat layout by parameter NAME (so a fresh process with the same model reloads it
    whatever the buffer order), and `reference_state_dict(model)` / `load_state_dict` speak the torch-optimizer
    format the reference writes into its checkpoints ({'state': {idx: {'step','exp_avg','exp_avg_sq'}},
    'param_groups': [...]}, run_PMR_ModCR.py:236)."""

    def __init__(self, flat_grads, names, learning_rate=1e-5, betas=(0.9, 0.999), adam_epsilon=1e-5,
                 weight_decay=0.0, t_total=1000, form="hf", scheduler="linear", warmup_steps=0):
        import modcr_hip as mh
        if form not in ("hf", "torch"):
            raise ValueError("FlatAdamW: form must be 'hf' or 'torch'")
        self.mh, self.fg, self.form = mh, flat_grads, form
        self.lr, self.betas, self.eps, self.wd, self.t_total = learning_rate, betas, adam_epsilon, weight_decay, t_total
        self.scheduler, self.warmup_steps = scheduler, warmup_steps
        self._lambda = lr_lambda(scheduler, warmup_steps, t_total)
        dev = flat_grads.flat.device
        self.flat_p = torch.zeros_like(flat_grads.flat)
        name_of = {id(p): n for p, n in zip(flat_grads.params, names)}
        self.segments, self.layout = [], []
        for p in flat_grads.order:                     # FlatGrads' layout (256-byte aligned starts, zero padding), increasing offset
            n = p.numel()
            off = flat_grads.offsets[id(p)]
            self.flat_p[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + n].view_as(p)
            scale = 0.1 if "seq_enc" in name_of[id(p)] else 1.0
            if self.segments and self.segments[-1][2] == scale:
                self.segments[-1][1] = off + n
            else:
                self.segments.append([off, off + n, scale])
            self.layout.append((name_of[id(p)], off, n, tuple(p.shape)))
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.sumsq = torch.zeros(1, dtype=torch.float32, device=dev)
        # workspace of the ordered norm (modcr_sumsq_f32_ordered): the clip coefficient is a pure function of the (all-reduced,
        # rank-identical) gradient buffer, so data-parallel replicas stay bit-identical
        self.sumsq_partials = torch.zeros(mh.sumsq_partials(), dtype=torch.float32, device=dev) if dev.type == "cuda" else None
        self.t = 0

    def lr_factor(self):
        return self._lambda(self.t)

    def step(self, max_grad_norm=1.0):
        factor = self.lr_factor()                      # LambdaLR: the step-t update uses the factor of t scheduler steps
        self.t += 1
        b1, b2 = self.betas
        self.sumsq.zero_()
        self.mh.sumsq_accumulate(self.fg.flat, self.sumsq, self.sumsq_partials)
        for s, e, scale in self.segments:
            self.mh.adamw_step(self.flat_p[s:e], self.fg.flat[s:e], self.exp_avg[s:e], self.exp_avg_sq[s:e], self.sumsq,
                               max_grad_norm, self.lr * scale * factor, b1, b2, self.eps, self.wd,
                               1.0 - b1 ** self.t, 1.0 - b2 ** self.t, form=self.form)
        for p in self.fg.params:                       # the kernels wrote through raw pointers: tell autograd / PackCache
            torch.autograd.graph.increment_version(p)

    def grad_norm(self):
        """global gradient norm seen by the last step() (device scalar -> host: a sync; logging only)"""
        return float(self.sumsq.sqrt().item())

    # ---- checkpointing -------------------------------------------------------------------------------------
    def state_dict(self):
        return {"format": "modcr_flat_adamw", "t": self.t, "exp_avg": self.exp_avg.detach().clone(),
                "exp_avg_sq": self.exp_avg_sq.detach().clone(), "layout": list(self.layout), "lr": self.lr,
                "betas": self.betas, "eps": self.eps, "weight_decay": self.wd, "t_total": self.t_total, "form": self.form,
                "scheduler": self.scheduler, "warmup_steps": self.warmup_steps}

    def reference_state_dict(self, model):
        """the same state in the format torch.optim.Optimizer.state_dict() gives the reference's AdamW"""
        g0, g1 = reference_param_order(model)
        index = {n: i for i, n in enumerate(g0 + g1)}
        state = {}
        if self.t > 0:
            for name, off, n, shape in self.layout:
                state[index[name]] = {"step": self.t, "exp_avg": self.exp_avg[off:off + n].view(shape).clone(),
                                      "exp_avg_sq": self.exp_avg_sq[off:off + n].view(shape).clone()}
        common = {"betas": self.betas, "eps": self.eps, "weight_decay": self.wd, "correct_bias": True}
        factor = self.lr_factor()
        groups = [dict(common, lr=self.lr * factor, initial_lr=self.lr, params=list(range(len(g0)))),
                  dict(common, lr=self.lr * 0.1 * factor, initial_lr=self.lr * 0.1, params=list(range(len(g0), len(g0) + len(g1))))]
        return {"state": state, "param_groups": groups}

    def load_state_dict(self, sd, model=None):
        """either format; a parameter the saved state does not know keeps zero moments (as a torch optimizer would)"""
        if "layout" in sd:
            mine = {name: (off, n) for name, off, n, _ in self.layout}
            unknown = [name for name, _, _, _ in sd["layout"] if name not in mine]
            if unknown:
                raise KeyError("optimizer state holds parameters this model does not train: %s" % unknown[:5])
            self.exp_avg.zero_(); self.exp_avg_sq.zero_()
            for name, off, n, _ in sd["layout"]:
                o2, n2 = mine[name]
                if n2 != n:
                    raise ValueError("optimizer state: size of %s is %d, expected %d" % (name, n, n2))
                self.exp_avg[o2:o2 + n].copy_(sd["exp_avg"][off:off + n])
                self.exp_avg_sq[o2:o2 + n].copy_(sd["exp_avg_sq"][off:off + n])
            self.t = int(sd["t"])
            return
        if "state" in sd and "param_groups" in sd:
            if model is None:
                raise ValueError("a torch-format optimizer state needs the model (parameter order)")
            g0, g1 = reference_param_order(model)
            names = g0 + g1
            mine = {name: (off, n, shape) for name, off, n, shape in self.layout}
            self.exp_avg.zero_(); self.exp_avg_sq.zero_()
            steps = set()
            for idx, st in sd["state"].items():
                name = names[int(idx)]
                if name not in mine:
                    raise KeyError("optimizer state holds %s, which this model does not train" % name)
                off, n, _ = mine[name]
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                steps.add(int(st["step"]))
            if len(steps) > 1:
                raise ValueError("optimizer state with different step counts per parameter (%s): the flat step shares one" % sorted(steps))
            self.t = steps.pop() if steps else 0
            return
        raise ValueError("unrecognised optimizer state (keys %s)" % sorted(sd.keys()))


class HFAdamW(torch.optim.Optimizer):
    """transformers.AdamW restated as a torch Optimizer (transformers 4.x optimization.py::AdamW.step, correct_bias=True):
    the non-fused route (gradient accumulation, where the reference clips every micro-batch: run_PMR_ModCR.py:216 vs :220).
    torch.optim.AdamW is NOT the same update (eps is added after the bias correction of the second moment there)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0, correct_bias=True):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, correct_bias=correct_bias))

    @torch.no_grad()
    def step(self, closure=None):
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                grad = p.grad
                state = self.state[p]
                if len(state) == 0:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p)
                    state["exp_avg_sq"] = torch.zeros_like(p)
                exp_avg, exp_avg_sq = state["exp_avg"], state["exp_avg_sq"]
                beta1, beta2 = group["betas"]
                state["step"] += 1
                exp_avg.mul_(beta1).add_(grad, alpha=1.0 - beta1)
                exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1.0 - beta2)
                denom = exp_avg_sq.sqrt().add_(group["eps"])
                step_size = group["lr"]
                if group["correct_bias"]:
                    step_size = step_size * (1.0 - beta2 ** state["step"]) ** 0.5 / (1.0 - beta1 ** state["step"])
                p.addcdiv_(exp_avg, denom, value=-step_size)
                if group["weight_decay"] > 0.0:
                    p.add_(p, alpha=-group["lr"] * group["weight_decay"])


    # ---- the reference's checkpoint layout (what FlatAdamW.reference_state_dict writes), so that resume files are
    # interchangeable between the fused and the non-fused route and across gradient-accumulation settings
    def reference_state_dict(self, model):
        g0, g1 = reference_param_order(model)
        index = {n: i for i, n in enumerate(g0 + g1)}
        name_of = {id(p): n for n, p in model.named_parameters()}
        state = {}
        for p, st in self.state.items():
            if st:
                state[index[name_of[id(p)]]] = {"step": int(st["step"]), "exp_avg": st["exp_avg"].detach().clone(),
                                                "exp_avg_sq": st["exp_avg_sq"].detach().clone()}
        grp = {("seq_enc" in name_of[id(g["params"][0])]): g for g in self.param_groups}
        def meta(g, lr_scale):
            src = g if g is not None else self.param_groups[0]
            d = {k: v for k, v in src.items() if k != "params"}
            if g is None:
                d["lr"] = src["lr"] * lr_scale
                d["initial_lr"] = src.get("initial_lr", src["lr"]) * lr_scale
            return d
        groups = [dict(meta(grp.get(False), 10.0 if False not in grp else 1.0), params=list(range(len(g0)))),
                  dict(meta(grp.get(True), 0.1), params=list(range(len(g0), len(g0) + len(g1))))]
        return {"state": state, "param_groups": groups}

    def load_reference_state_dict(self, sd, model):
        """moments and step counts by parameter NAME from the reference layout; learning rates stay this run's (the scheduler
        state restores them)"""
        g0, g1 = reference_param_order(model)
        names = g0 + g1
        params = dict(model.named_parameters())
        mine = {id(p) for g in self.param_groups for p in g["params"]}
        for idx, st in sd["state"].items():
            p = params[names[int(idx)]]
            if id(p) not in mine:
                raise KeyError("optimizer state holds %s, which this run does not train" % names[int(idx)])
            self.state[p] = {"step": int(st["step"]), "exp_avg": st["exp_avg"].to(p.device, p.dtype).clone(),
                             "exp_avg_sq": st["exp_avg_sq"].to(p.device, p.dtype).clone()}


def make_optimizer(model, names, learning_rate=1e-5, adam_epsilon=1e-5, t_total=1000, scheduler="linear", warmup_steps=0):
    """run_PMR_ModCR.py:127-145: transformers.AdamW (weight_decay 0), 'seq_enc' group at lr*0.1 (empty here unless the
    encoders are trained), linear / constant schedule with warm-up."""
    params = dict(model.named_parameters())
    groups = [{"params": [params[n] for n in names if "seq_enc" not in n], "lr": learning_rate},
              {"params": [params[n] for n in names if "seq_enc" in n], "lr": learning_rate * 0.1}]
    groups = [g for g in groups if g["params"]]
    opt = HFAdamW(groups, lr=learning_rate, eps=adam_epsilon, weight_decay=0.0)
    sched = torch.optim.lr_scheduler.LambdaLR(opt, lr_lambda(scheduler, warmup_steps, t_total))
    return opt, sched


def pack_gather_index(gather_index, t):
    """Host-side packing of the ragged `gather_index` list (Data/VCRChunkAlign.py:666-670) into the
    int32 [N,T] chunk-id tensor the attention kernel reads (row t of sequence n = chunk of text token
    t, -1 = leave the query alone): one H2D copy per batch instead of N tiny ones per forward."""
    import numpy as np
    cid = np.full((len(gather_index), t), -1, np.int32)
    for i, g in enumerate(gather_index):
        g = g.detach().cpu().numpy() if torch.is_tensor(g) else np.asarray(g)
        cid[i, 1:1 + min(len(g), t - 1)] = g[:t - 1]
    return torch.from_numpy(cid)


def batch_to_device(batch, device):
    out = {}
    if isinstance(batch.get("gather_index"), list) and torch.is_tensor(batch.get("input_ids")):
        batch = dict(batch, gather_index=pack_gather_index(batch["gather_index"], batch["input_ids"].shape[1]))
    for k, v in batch.items():
        if torch.is_tensor(v):
            out[k] = v.to(device, non_blocking=True)
        elif isinstance(v, list) and v and torch.is_tensor(v[0]):
            out[k] = [t.to(device, non_blocking=True) for t in v]
        else:
            out[k] = v
    return out


def forward_inputs(batch):
    """the kwargs dict of run_PMR_ModCR.py:189-200"""
    return {'input_ids': batch['input_ids'], 'image': batch['image'], 'text': batch['text'],
            'roberta_input_ids': batch['r_input_ids'], 'roberta_token_type_ids': batch['r_token_type_ids'],
            'roberta_attention_mask': batch['r_attention_mask'], 'token_type_ids': batch['token_type_ids'],
            'input_mask': batch['input_mask'], 'img_feat': batch['img_feat'], 'label': batch['label'],
            'gather_index': batch['gather_index'], 'offsets': batch['offsets'],
            'chunk_attention_mask': batch['chunk_attention_mask'], 'align_pos': batch['align_pos'],
            'total_label': batch['total_label']}


def micro_step(model, batch, optimizer, scheduler, flat, world_size=1, max_grad_norm=1.0, accumulation_steps=1, last=True):
    """One micro-batch of train() (run_PMR_ModCR.py:188-227): forward, loss / accumulation_steps (:212-213), backward,
    clip_grad_norm_ on the ACCUMULATED gradient after every micro-batch (:216 sits before the `if (step + 1) % accumulation`
    of :220), and on the last micro-batch of a window optimizer.step(), scheduler.step(), zero_grad (:224-227).
    Returns (the scaled loss as the reference logs it, logits).

    Every micro-step is all-reduced, so that the per-micro-step clip (the reference is a single process, it never wraps the
    model in DistributedDataParallel) acts on the rank-mean accumulated gradient -- what one process with a world-size-times
    larger batch would clip.  The buffer already holds the earlier micro-steps' reduced (identical on every rank) sum:
    SUM / world of (that + the local new gradient) leaves it as it is and adds the mean of the new one.
    The fused optimizer (FlatAdamW: norm + clip + AdamW + schedule as two kernels) serves accumulation_steps == 1, where the
    one clip is part of its step; with accumulation the per-tensor route (make_optimizer) clips here."""
    flat.new_graph()                # use counts of a forward whose backward never ran (an eval pass with grad mode on) are not this graph's
    outputs = model(**forward_inputs(batch))
    loss = outputs[0]
    if accumulation_steps > 1:
        loss = loss / accumulation_steps
    fused = isinstance(optimizer, FlatAdamW)
    if fused and accumulation_steps > 1:
        raise ValueError("FlatAdamW clips once per optimizer step: gradient accumulation takes make_optimizer()'s per-tensor route")
    flat.begin(world_size)          # bucketed all-reduce launched from gradient hooks / sink reports during backward
    loss.backward()
    flat.finish(world_size)
    if not fused:
        torch.nn.utils.clip_grad_norm_(flat.params, max_grad_norm)
    if last:
        if fused:
            optimizer.step(max_grad_norm)                   # norm + clip + AdamW + schedule: two kernels
        else:
            optimizer.step()
            scheduler.step()
        flat.zero()                 # model.zero_grad() with the flat buffer kept in place
    return loss, outputs[2]


def train_step(model, batch, optimizer, scheduler, flat, world_size=1, max_grad_norm=1.0):
    """One optimisation step of train() (run_PMR_ModCR.py:188-227) with gradient_accumulation_steps = 1."""
    return micro_step(model, batch, optimizer, scheduler, flat, world_size, max_grad_norm, 1, True)
