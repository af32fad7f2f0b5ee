#!/usr/bin/env python
"""run_PMR_ModCR.py -- entry point of the MI355X-native ModCR hot path (PMR, 4-way multiple choice).

Mirrors the reference's run_PMR_ModCR.py (argparse flags :486-681, main :451-925, train :115-240,
eval :243-280, test :283-353) for the path this build covers: model assembly
(BertImgModel -> SeqBertImgModel -> ChunkAlign_CLS_enc4_align_ensemble -> Abstract_Specific),
AdamW + linear decay, clip_grad_norm_ every micro-step, validation every --valid_steps, best-val
checkpoint {'net','optimizer','epoch'}.  The reference's pickled PMR features, tokenizers, CLIP and
RoBERTa weights are not in its tree (.MISSING_LARGE_BLOBS): data comes from Data/synthetic.py,
which honours the same batch contract, and the RoBERTa body is the stand-in of
modeling/roberta_prefix.py unless --roberta_body large selects the 24-layer prefix RoBERTa-large on the HIP kernels.

One process per GPU.  Multi-GPU: `python -m torch.distributed.run --nproc-per-node N
run_PMR_ModCR.py ...` -- pure data parallel, one RCCL all-reduce of the trainable gradients per
step.  There is no CPU path: the HIP library must be built (make -C csrc).
"""
import argparse
import json
import logging
import os
import sys
import time

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, SequentialSampler
from torch.utils.data.distributed import DistributedSampler

HERE = os.path.dirname(os.path.abspath(__file__))
if HERE not in sys.path:
    sys.path.insert(0, HERE)

import modcr_hip  # noqa: E402
from Data.synthetic import SyntheticPMRDataset  # noqa: E402
from modeling import train_utils as tu  # noqa: E402

logger = logging.getLogger("run_PMR_ModCR")


def make_data_loader(args, dataset, is_distributed=True, is_train=True):
    """run_PMR_ModCR.py:40-48"""
    if is_train:
        batch_size = args.per_gpu_train_batch_size
        sampler = DistributedSampler(dataset, shuffle=True) if is_distributed else torch.utils.data.RandomSampler(dataset)
    else:
        batch_size = args.per_gpu_eval_batch_size
        sampler = SequentialSampler(dataset)
    return DataLoader(dataset, batch_size=batch_size, sampler=sampler, drop_last=is_train,
                      collate_fn=dataset.SNLIGPT_gen_collate, num_workers=0)


def evaluate(args, dataloader, model):
    """eval(): argmax(outputs[2]) vs argmax(label.reshape(-1, 4)), running accuracy (:243-280)."""
    model.eval()
    acc = cnt = 0
    with torch.no_grad():
        for batch in dataloader:
            batch = tu.batch_to_device(batch, args.device)
            inputs = tu.forward_inputs(batch)
            inputs["align_pos"] = inputs["total_label"] = None       # eval() does not pass them (:253-263)
            logits = model(**inputs)[2]
            pred = torch.argmax(logits, -1)
            label = torch.argmax(batch["label"].reshape(-1, 4), -1)
            acc += int((pred == label).sum().item())
            cnt += int(label.numel())
    return acc / max(cnt, 1)


def test(args, dataloader, model):
    """test(): per-example predictions as JSON lines (:283-353)."""
    model.eval()
    rows = []
    with torch.no_grad():
        for step, batch in enumerate(dataloader):
            batch = tu.batch_to_device(batch, args.device)
            inputs = tu.forward_inputs(batch)
            inputs["label"] = inputs["align_pos"] = inputs["total_label"] = None
            logits = model(**inputs)[2]
            for i, row in enumerate(logits.float().cpu().tolist()):
                rows.append({"index": step * args.per_gpu_eval_batch_size + i,
                             "prediction": int(max(range(4), key=lambda c: row[c])), "logits": row})
    out = os.path.join(args.output_dir, "test_predictions.jsonl")
    with open(out, "w") as f:
        for r in rows:
            f.write(json.dumps(r) + "\n")
    return out


CKPT_TAG = "Multi-View-Reasoning-Prefix-tuning_LV_3_LA_7"      # run_PMR_ModCR.py:236


def save_resume_state(args, model, optimizer, scheduler, global_step, epoch):
    """model.pth / optimizer.pth / scheduler.pth in <output_dir>/last/: the three files the reference's resume path reads
    from --eval_model_dir when --global_step > 0 (run_PMR_ModCR.py:146-156; nothing in the reference writes them)."""
    d = os.path.join(args.output_dir, "last")
    os.makedirs(d, exist_ok=True)
    torch.save(model.state_dict(), os.path.join(d, "model.pth"))
    # always the torch-optimizer layout the reference's AdamW writes: both optimizers (fused flat / HFAdamW) load it, whatever the
    # gradient-accumulation setting of the run that resumes
    opt_sd = optimizer.reference_state_dict(model) if hasattr(optimizer, "reference_state_dict") else optimizer.state_dict()
    torch.save(opt_sd, os.path.join(d, "optimizer.pth"))
    sched = scheduler.state_dict() if scheduler is not None else {"last_epoch": optimizer.t, "scheduler": args.scheduler,
                                                                  "warmup_steps": args.warmup_steps}
    torch.save(dict(sched, global_step=global_step, epoch=epoch), os.path.join(d, "scheduler.pth"))
    return d


def train(args, train_dataloader, val_dataloader, model):
    names = tu.trainable_parameters(model)
    pdict = dict(model.named_parameters())
    for k, p in pdict.items():
        p.requires_grad_(k in names)
    flat = tu.FlatGrads([pdict[k] for k in names], args.device, names=names,
                        comm_dtype=torch.bfloat16 if getattr(args, "modcr_grad_comm", "fp32") == "bf16" else None)
    steps_per_epoch = max(1, len(train_dataloader) // args.gradient_accumulation_steps)
    if args.max_steps > 0:                                          # run_PMR_ModCR.py:118-124
        t_total = args.max_steps
        args.num_train_epochs = args.max_steps // steps_per_epoch + 1
    else:
        t_total = steps_per_epoch * args.num_train_epochs
    # gradient_accumulation_steps == 1 (the PMR default): norm + clip + AdamW + schedule as two kernels over the flat
    # buffers; with accumulation the reference clips every micro-batch (:216 vs :220), which takes the per-tensor route
    fused = args.gradient_accumulation_steps == 1
    if fused:
        optimizer, scheduler = tu.FlatAdamW(flat, names, args.learning_rate, adam_epsilon=args.adam_epsilon, t_total=t_total,
                                            scheduler=args.scheduler, warmup_steps=args.warmup_steps), None
    else:
        optimizer, scheduler = tu.make_optimizer(model, names, args.learning_rate, args.adam_epsilon, t_total,
                                                 scheduler=args.scheduler, warmup_steps=args.warmup_steps)
    if args.global_step > 0:                                        # resume (:146-156)
        d = args.eval_model_dir
        if not os.path.isdir(d):
            raise SystemExit("--global_step %d: resume needs a directory with model.pth / optimizer.pth / scheduler.pth, got %r"
                             % (args.global_step, d))
        model.load_state_dict(torch.load(os.path.join(d, "model.pth"), map_location="cpu"))       # strict, as the reference
        osd = torch.load(os.path.join(d, "optimizer.pth"), map_location="cpu", weights_only=False)
        ssd = torch.load(os.path.join(d, "scheduler.pth"), map_location="cpu", weights_only=False)
        if fused:
            optimizer.load_state_dict(osd, model=model)
            optimizer.t = int(ssd.get("last_epoch", optimizer.t))
        else:
            optimizer.load_reference_state_dict(osd, model)
            # either route's scheduler.pth: only the step count matters (the schedule itself is this run's command line)
            scheduler.last_epoch = int(ssd.get("last_epoch", 0))
            for g, lam, base in zip(optimizer.param_groups, scheduler.lr_lambdas, scheduler.base_lrs):
                g["lr"] = base * lam(scheduler.last_epoch)
            scheduler._last_lr = [g["lr"] for g in optimizer.param_groups]
        logger.info("  Resume from %s", d)
    logger.info("***** Running training *****  steps/epoch = %d, epochs = %d, total optimization steps = %d, trainable tensors = %d",
                len(train_dataloader), args.num_train_epochs, t_total, len(names))
    global_step, best_acc = args.global_step, 0.0
    model.train()
    for epoch in range(int(args.num_train_epochs)):
        if isinstance(train_dataloader.sampler, DistributedSampler):
            train_dataloader.sampler.set_epoch(epoch)
        global_loss, new_step, t0 = 0.0, 0, time.time()
        for step, batch in enumerate(train_dataloader):
            batch = tu.batch_to_device(batch, args.device)
            last = (step + 1) % args.gradient_accumulation_steps == 0
            # forward, loss / accumulation, bucketed all-reduce under backward, per-micro-batch clip, optimizer + schedule on the
            # last micro-batch of a window: modeling/train_utils.py::micro_step (run_PMR_ModCR.py:201-227)
            loss, _ = tu.micro_step(model, batch, optimizer, scheduler, flat, args.world_size, args.max_grad_norm,
                                    args.gradient_accumulation_steps, last)
            global_loss += loss.item()
            if last:
                new_step += 1
                global_step += 1
                if args.logging_steps and global_step % args.logging_steps == 0 and args.rank == 0:
                    logger.info("Epoch %d step %d loss %.4f (%.1f examples/s)", epoch + 1, global_step,
                                global_loss / new_step,
                                new_step * args.per_gpu_train_batch_size * args.gradient_accumulation_steps * args.world_size / (time.time() - t0))
                if epoch >= args.epoch_begin - 1 and global_step % args.valid_steps == 0:
                    acc = evaluate(args, val_dataloader, model)
                    logger.info("when epoch %d, the accuracy is %s", epoch + 1, acc)
                    if acc > best_acc and args.rank == 0:
                        best_acc = acc
                        # {'net','optimizer','epoch'} (:236); the optimizer state in the torch format the reference's AdamW writes
                        opt_sd = optimizer.reference_state_dict(model) if fused else optimizer.state_dict()
                        state = {"net": model.state_dict(), "optimizer": opt_sd, "epoch": epoch}
                        path = os.path.join(args.output_dir, "%s-%d-%s-%d.pth" % (CKPT_TAG, epoch + 1, acc, global_step))
                        torch.save(state, path)
                        args.last_checkpoint = path
                    if args.rank == 0:          # the LAST state, on every validation (a plateaued run still has a recent resume point)
                        save_resume_state(args, model, optimizer, scheduler, global_step, epoch)
                    model.train()
                if args.max_steps > 0 and global_step >= args.max_steps:
                    return global_step, global_loss / max(new_step, 1)
    return global_step, global_loss / max(new_step, 1)


# The reference's command line (run_PMR_ModCR.py:486-681): every flag is accepted with the reference's default, so a
# reference command line parses here unchanged.  (name, type, default); flags this path does not use are listed once at
# start-up ("parsed, unused") -- the reference itself ignores most of them (they are inherited from Oscar captioning).
_PMR = "pmr_data/"
REFERENCE_FLAGS = [
    ("roberta_file_train", str, _PMR + "train_CALeC.pkl"), ("roberta_file_dev", str, _PMR + "val_CALeC.pkl"),
    ("roberta_file_test", str, _PMR + "test_CALeC.pkl"),
    ("clip_file_train", str, _PMR + "clip_data/train_p_ori-clip.jsonl"), ("clip_file_dev", str, _PMR + "clip_data/val_p_ori-clip.jsonl"),
    ("clip_file_test", str, _PMR + "clip_data/test_p_ori-clip.jsonl"),
    ("vcr_example_file_train", str, _PMR + "ex_feature/train_CALeC_ori-o.pkl"), ("vcr_example_file_dev", str, _PMR + "ex_feature/val_CALeC_ori-o.pkl"),
    ("vcr_example_file_test", str, _PMR + "ex_feature/test_CALeC_ori-o.pkl"),
    ("vcr_feat_file_train", str, _PMR + "image_feature/train_feat_m.pkl"), ("vcr_feat_file_dev", str, _PMR + "image_feature/val_feat_m.pkl"),
    ("vcr_feat_file_test", str, _PMR + "image_feature/test_feat_m.pkl"),
    ("vcr_chunk_mask_train", str, _PMR + "ChunkMaskTrain_v4_without_premise.pkl"), ("vcr_chunk_mask_dev", str, _PMR + "ChunkMaskVal_v4_without_premise.pkl"),
    ("vcr_chunk_mask_test", str, _PMR + "ChunkMaskTest_v4_without_premise.pkl"),
    ("num_gpus", int, 1), ("train_yaml", str, "train.yaml"), ("test_yaml", str, "test.yaml"), ("val_yaml", str, "val.yaml"),
    ("gpt_model_name_or_path", str, "./GPT2"),
    ("model_name_or_path", str, "./Oscar/image-captioning/pretrained_base/checkpoint-2000000/"),
    ("seq_model_name_or_path", str, "./Oscar/image-captioning/pretrained_base/checkpoint-2000000/"),
    ("seq_pretrain_model_dir", str, "./local_transformers/checkpoint-6-2625-acc-0.8164/checkpoint-6-2625-acc-0.8164/"),
    ("output_dir", str, "./output/checkpoint/Tu/"), ("loss_type", str, "sfmx"), ("config_name", str, ""), ("tokenizer_name", str, ""),
    ("max_seq_length", int, 140), ("max_hypo_len", int, 50), ("mask_prob", float, 0.0), ("max_masked_tokens", int, 3),
    ("drop_out", float, 0.3), ("max_img_seq_length", int, 150), ("img_feature_dim", int, 2054), ("img_feature_type", str, "frcnn"),
    ("label_smoothing", float, 0), ("drop_worst_ratio", float, 0), ("drop_worst_after", int, 0),
    ("per_gpu_train_batch_size", int, 16), ("per_gpu_eval_batch_size", int, 4), ("output_mode", str, "classification"), ("num_labels", int, 2),
    ("gradient_accumulation_steps", int, 1), ("learning_rate", float, 1e-5), ("weight_decay", float, 0.05), ("adam_epsilon", float, 1e-5),
    ("max_grad_norm", float, 1.0), ("warmup_steps", int, 0), ("scheduler", str, "linear"), ("num_workers", int, 4),
    ("num_train_epochs", int, 30), ("max_steps", int, -1), ("logging_steps", int, 200), ("save_steps", int, 1000),
    ("local_rank", int, 0), ("seed", int, 88), ("sc_train_sample_n", int, 2), ("sc_baseline_type", str, "greedy"), ("beam_size", int, 5),
    ("cider_cached_tokens", str, "coco-train-words.p"),
    ("eval_model_dir", str, "output/checkpoint/Tu/Multi-View-Reasoning-Prefix-tuning_len5-6-0.8491547464239272-4500.pth"),
    ("max_gen_length", int, 40), ("num_return_sequences", int, 1), ("num_beams", int, 1), ("num_keep_best", int, 1),
    ("temperature", float, 1), ("top_k", int, 0), ("top_p", float, 1), ("repetition_penalty", int, 1), ("length_penalty", int, 1),
    ("min_constraints_to_satisfy", int, 2), ("epoch_begin", int, 2), ("valid_steps", int, 400), ("result_dir", str, "output/results/"),
    ("global_step", int, 0), ("example_index", int, None),
]
REFERENCE_SWITCHES = ["do_train", "do_test", "do_eval", "add_residual", "add_local_residual", "wo_gate", "do_lower_case", "add_od_labels",
                      "tie_weights", "freeze_embedding", "no_cuda", "scst", "output_hidden_states", "compressed_db", "use_cbs"]
# what this path reads; everything else is parsed and ignored (as the reference ignores it on the ModCR path)
USED_FLAGS = {"model_name_or_path", "seq_model_name_or_path", "seq_pretrain_model_dir", "output_dir", "max_hypo_len", "drop_out",
              "max_img_seq_length", "img_feature_dim", "per_gpu_train_batch_size", "per_gpu_eval_batch_size", "gradient_accumulation_steps",
              "learning_rate", "adam_epsilon", "max_grad_norm", "warmup_steps", "scheduler", "num_train_epochs", "max_steps",
              "logging_steps", "local_rank", "seed", "eval_model_dir", "epoch_begin", "valid_steps", "global_step", "do_train", "do_test",
              "do_eval", "add_residual", "add_local_residual", "config_name"}


MODEL_HOOKS = []       # callables run on the freshly built model (before checkpoints load and before the optimizer sees it)


def get_args(argv=None):
    p = argparse.ArgumentParser()
    for name, typ, default in REFERENCE_FLAGS:
        if name == "local_rank":
            default = int(os.environ.get("LOCAL_RANK", default))
        p.add_argument("--" + name, default=default, type=typ)
    for name in REFERENCE_SWITCHES:
        p.add_argument("--" + name, action="store_true")
    # ---- this build (not in the reference) ----
    p.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--modcr_last_layer_rows", action="store_true",
                   help="(this build) the frozen encoders' last layers run BertSelfOutput / BertIntermediate / BertOutput only over the rows ModCR "
                        "reads (text rows; the [CLS] row of the image-only pass): same loss, logits and gradients, ~3 %% less time per step")
    p.add_argument("--modcr_grad_comm", default="fp32", choices=["fp32", "bf16"],
                   help="(this build, N > 1) type the gradient buckets cross xGMI in: fp32 (default, what one process would compute) or bf16 "
                        "(half the bytes: 0.83 instead of 1.66 GB per step with the RoBERTa body trainable; the buffer, clip and AdamW stay fp32)")
    p.add_argument("--modcr_backward_memory", default="keep", choices=["keep", "recompute", "auto"],
                   help="(this build) what trainable layers keep for their backward: keep = Q|K|V images + bf16 GELU input per layer (fastest, "
                        "~1 GB per layer at 128 examples); recompute = neither (the attention / FFN backward recompute them); auto = by free memory")
    p.add_argument("--roberta_body", default="standin", choices=["standin", "large"],
                   help="large = the 24-layer prefix RoBERTa-large on the HIP kernels, trainable (run_PMR_ModCR.py:772-781; "
                        "random init: local_transformers/roberta-large is not in the reference tree)")
    p.add_argument("--attention_probs_dropout_prob", default=None, type=float,
                   help="default: the value in <model_name_or_path>/config.json, else 0.1 (the BERT / Oscar checkpoints' value); live "
                        "under model.train() like every dropout of the path (modeling_bert.py:62,69)")
    p.add_argument("--hidden_size", default=768, type=int, help="encoder width (1024 = the Oscar-large shape class of BASELINE configs[4])")
    p.add_argument("--num_hidden_layers", default=12, type=int)
    p.add_argument("--num_attention_heads", default=None, type=int, help="default hidden_size / 64")
    p.add_argument("--synthetic_text_len", default=80, type=int, help="T of the synthetic batches (SURVEY 8d: PMR 80, VCR 194)")
    p.add_argument("--synthetic_regions", default=100, type=int, help="R of the synthetic batches (PMR 100, VCR 36)")
    p.add_argument("--synthetic_train_examples", default=4096, type=int)
    p.add_argument("--synthetic_val_examples", default=256, type=int)
    p.add_argument("--random_init", action="store_true",
                   help="allow --do_eval / --do_test without a checkpoint (random weights; smoke runs only)")
    p.add_argument("--device", default="cuda", type=str)
    args = p.parse_args(argv)
    given = set()
    for a in (sys.argv[1:] if argv is None else argv):
        if a.startswith("--"):
            given.add(a[2:].split("=")[0])
    args.ignored_flags = sorted(n for n in given if n in {f[0] for f in REFERENCE_FLAGS} | set(REFERENCE_SWITCHES) and n not in USED_FLAGS)
    return args


def load_checkpoint_net(model, path):
    """the reference's --do_test load (run_PMR_ModCR.py:805-808): strict load of ck['net'] from the .pth file"""
    ck = torch.load(path, map_location="cpu", weights_only=False)
    if not isinstance(ck, dict) or "net" not in ck:
        raise SystemExit("%s is not a ModCR checkpoint ({'net','optimizer','epoch'})" % path)
    model.load_state_dict(ck["net"])            # strict
    return ck


def load_pretrained(args, model):
    """Oscar / seq_enc / cold-start weights when the files exist (run_PMR_ModCR.py:727-764,820-832); every partial load
    reports what it did not find instead of hiding it behind strict=False."""
    def report(what, res):
        if res.missing_keys or res.unexpected_keys:
            logger.warning("%s: %d missing keys (e.g. %s), %d unexpected keys (e.g. %s)", what, len(res.missing_keys), res.missing_keys[:3],
                           len(res.unexpected_keys), res.unexpected_keys[:3])
    f = os.path.join(args.model_name_or_path or "", "pytorch_model.bin")
    if os.path.isfile(f):
        sd = torch.load(f, map_location="cpu")
        report("Oscar checkpoint " + f, model.calec.global_enc.load_state_dict({k[5:] if k.startswith("bert.") else k: v for k, v in sd.items()}, strict=False))
    else:
        logger.info("no Oscar checkpoint at %s: global_enc keeps its random initialisation", f)
    f = os.path.join(args.seq_pretrain_model_dir or "", "model.pth")
    if os.path.isfile(f):
        sd = torch.load(f, map_location="cpu")
        sd = sd.get("net", sd)
        model.calec.seq_enc.load_state_dict({".".join(k.split(".")[1:]): v for k, v in sd.items() if "seq_enc" in k})      # strict (:759-763)
        logger.info("load pretrained ChunkAlign from %s", args.seq_pretrain_model_dir)
    else:
        logger.info("no phrase-level aligner checkpoint at %s: seq_enc keeps its random initialisation", f)
    f = os.path.join(args.output_dir, "Multi-View-Reasoning-cold-start-1-0.19700910273081926-755.pth")
    if os.path.isfile(f):                                           # :820-832
        params = torch.load(f, map_location="cpu", weights_only=False)["net"]
        params = {k: v for k, v in params.items() if not any(t in k for t in ("mapping_network_vision.", "classifier.", "mapping_network_alignment."))}
        report("cold-start checkpoint " + f, model.load_state_dict(params, strict=False))


def main(argv=None):
    args = get_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(name)s %(message)s")
    modcr_hip.lib()                                    # fail loudly without the HIP library
    if not str(args.device).startswith("cuda") or not torch.cuda.is_available():
        raise SystemExit("run_PMR_ModCR.py needs an MI355X: the ModCR hot path has no CPU fallback "
                         "(the CPU restatement lives under oracle/ and is test infrastructure only)")
    if args.ignored_flags:
        logger.info("parsed, unused on the ModCR path (as in the reference): %s", " ".join("--" + f for f in args.ignored_flags))
    assert args.valid_steps % args.gradient_accumulation_steps == 0          # run_PMR_ModCR.py:694
    args.world_size = int(os.environ.get("WORLD_SIZE", 1))
    args.rank = int(os.environ.get("RANK", 0))
    args.distributed = args.world_size > 1
    torch.cuda.set_device(args.local_rank)
    args.device = torch.device("cuda", args.local_rank)
    if args.distributed:                               # run_PMR_ModCR.py:438-448 (nccl == RCCL on ROCm)
        tu.cap_host_threads(int(os.environ.get("LOCAL_WORLD_SIZE", args.world_size)))      # the node's CPU quota is shared by its ranks
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://")
        dist.barrier()
    os.makedirs(args.output_dir, exist_ok=True)
    torch.manual_seed(args.seed)

    attn_p = args.attention_probs_dropout_prob
    if attn_p is None:
        attn_p = 0.1
        cfg_file = os.path.join(args.config_name if args.config_name else (args.model_name_or_path or ""), "config.json")
        if os.path.isfile(cfg_file):
            attn_p = float(json.load(open(cfg_file)).get("attention_probs_dropout_prob", 0.1))
    heads = args.num_attention_heads or args.hidden_size // 64
    model = tu.build_model(args.device, dtype=args.dtype, seed=args.seed, roberta_body=args.roberta_body,
                           hidden_dropout_prob=args.drop_out, attention_probs_dropout_prob=attn_p,
                           roberta_hidden_dropout_prob=0.1 if args.roberta_body == "large" else 0.0,
                           hidden_size=args.hidden_size, num_hidden_layers=args.num_hidden_layers, num_attention_heads=heads,
                           max_hypo=args.max_hypo_len, add_residual=args.add_residual, add_local_residual=args.add_local_residual,
                           **({"modcr_last_layer_rows": True} if args.modcr_last_layer_rows else {}))
    for hook in MODEL_HOOKS:                           # e.g. run_vcr_ModCR.py's RoBERTa freeze (run_vcr_ModCR.py:781-787)
        hook(model)
    import modcr_hip as mh
    from modeling import hip_layers
    n_seq = 4 * max(1, args.per_gpu_train_batch_size)
    # the estimate of `auto` comes from the BUILT model: every group of trainable layers with its own (layers, sequences, length, width) --
    # 12 + 12 Oscar layers over text + regions with --train_encoders, 24 RoBERTa-large layers over its tokens + prefix
    groups = hip_layers.backward_memory_groups(model, n_seq, args.synthetic_text_len, args.synthetic_regions)
    chosen = hip_layers.configure_backward_memory(args.modcr_backward_memory, device=args.device if str(args.device).startswith("cuda") else None,
                                                  groups=groups)
    if args.modcr_backward_memory != "keep":
        logger.info("modcr_backward_memory=%s -> %s", args.modcr_backward_memory, chosen)
    mh.DROPOUT.manual_seed(args.seed + 7919 * getattr(args, "rank", 0))   # different masks per rank (different data anyway)
    if args.do_test or (args.do_eval and not args.do_train):
        if os.path.isfile(args.eval_model_dir):
            load_checkpoint_net(model, args.eval_model_dir)
            logger.info("loaded %s", args.eval_model_dir)
        elif not args.random_init:
            raise SystemExit("--do_test / --do_eval: no checkpoint file at --eval_model_dir %r (the reference loads ck['net'] from it, "
                             "run_PMR_ModCR.py:805-808); --random_init evaluates untrained weights on purpose" % args.eval_model_dir)
    else:
        load_pretrained(args, model)
    if args.distributed:                               # every rank starts from rank 0's weights
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, 0)

    T, R = args.synthetic_text_len, args.synthetic_regions
    val_ds = SyntheticPMRDataset(args.synthetic_val_examples, T=T, R=R, seed=4321, img_dim=args.img_feature_dim)
    val_dl = make_data_loader(args, val_ds, is_distributed=False, is_train=False)
    if args.do_train:
        train_ds = SyntheticPMRDataset(args.synthetic_train_examples, T=T, R=R, seed=1234, img_dim=args.img_feature_dim)
        train_dl = make_data_loader(args, train_ds, is_distributed=args.distributed, is_train=True)
        step, loss = train(args, train_dl, val_dl, model)
        logger.info("Training done: total_step = %s, avg loss = %s", step, loss)
    if args.do_eval:
        logger.info("validation accuracy %.4f", evaluate(args, val_dl, model))
    if args.do_test:
        logger.info("wrote %s", test(args, val_dl, model))
    if args.distributed:
        dist.barrier()
        dist.destroy_process_group()
    return args


if __name__ == "__main__":
    main()
