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


def train(args, train_dataloader, val_dataloader, model):
    names = tu.trainable_parameters(model)
    pdict = dict(model.named_parameters())
    for k, p in pdict.items():
        p.requires_grad_(k in names)
    flat = tu.FlatGrads([pdict[k] for k in names], args.device)
    t_total = len(train_dataloader) // args.gradient_accumulation_steps * args.num_train_epochs
    # gradient_accumulation_steps == 1 (the PMR default): norm + clip + AdamW + linear decay as two kernels over the
    # flat buffers; with accumulation the reference clips every micro-batch (:216 vs :220), which needs torch's path
    fused = args.gradient_accumulation_steps == 1
    if fused:
        optimizer, scheduler = tu.FlatAdamW(flat, names, args.learning_rate, adam_epsilon=args.adam_epsilon, t_total=t_total), None
    else:
        optimizer, scheduler = tu.make_optimizer(model, names, args.learning_rate, args.adam_epsilon, t_total)
    if args.global_step > 0 and args.eval_model_dir:               # resume (:146-156)
        optimizer.load_state_dict(torch.load(os.path.join(args.eval_model_dir, "optimizer.pth"), map_location="cpu"))
        if scheduler is not None:
            scheduler.load_state_dict(torch.load(os.path.join(args.eval_model_dir, "scheduler.pth"), map_location="cpu"))
    logger.info("***** Running training *****  steps/epoch = %d, epochs = %d, trainable tensors = %d",
                len(train_dataloader), args.num_train_epochs, len(names))
    global_step, best_acc = args.global_step, 0.0
    model.train()
    for epoch in range(int(args.num_train_epochs)):
        if isinstance(train_dataloader.sampler, DistributedSampler):
            train_dataloader.sampler.set_epoch(epoch)
        global_loss, new_step, t0 = 0.0, 0, time.time()
        for step, batch in enumerate(train_dataloader):
            batch = tu.batch_to_device(batch, args.device)
            loss = model(**tu.forward_inputs(batch))[0]
            if args.gradient_accumulation_steps > 1:
                loss = loss / args.gradient_accumulation_steps
            last = (step + 1) % args.gradient_accumulation_steps == 0
            if last:
                flat.begin(args.world_size)     # bucketed all-reduce launched from gradient hooks during backward
            loss.backward()
            if last:
                flat.finish(args.world_size)
            if not fused:
                torch.nn.utils.clip_grad_norm_(flat.params, args.max_grad_norm)
            global_loss += loss.item()
            if last:
                new_step += 1
                global_step += 1
                if fused:
                    optimizer.step(args.max_grad_norm)
                else:
                    optimizer.step()
                    scheduler.step()
                flat.zero()
                if args.logging_steps and global_step % args.logging_steps == 0 and args.rank == 0:
                    logger.info("Epoch %d step %d loss %.4f (%.1f examples/s)", epoch + 1, global_step,
                                global_loss / new_step,
                                new_step * args.per_gpu_train_batch_size * args.world_size / (time.time() - t0))
                if epoch >= args.epoch_begin - 1 and global_step % args.valid_steps == 0:
                    acc = evaluate(args, val_dataloader, model)
                    logger.info("when epoch %d, the accuracy is %.4f", epoch + 1, acc)
                    if acc > best_acc and args.rank == 0:
                        best_acc = acc
                        state = {"net": model.state_dict(), "optimizer": optimizer.state_dict(), "epoch": epoch}
                        torch.save(state, os.path.join(args.output_dir, "Multi-View-Reasoning-Prefix-tuning_LV_3_LA_7-%d-%s-%d.pth"
                                                       % (epoch + 1, acc, global_step)))
                    model.train()
                if args.max_steps and global_step >= args.max_steps:
                    return global_step, global_loss / max(new_step, 1)
    return global_step, global_loss / max(new_step, 1)


def get_args(argv=None):
    p = argparse.ArgumentParser()
    # the flags of the reference that matter on this path (run_PMR_ModCR.py:486-681); defaults kept
    p.add_argument("--model_name_or_path", default=None, type=str, help="Oscar checkpoint dir (optional)")
    p.add_argument("--seq_model_name_or_path", default=None, type=str, help="state dict with 'seq_enc.*' keys (optional)")
    p.add_argument("--eval_model_dir", default="", type=str)
    p.add_argument("--output_dir", default="output/", type=str)
    p.add_argument("--do_train", action="store_true")
    p.add_argument("--do_test", action="store_true")
    p.add_argument("--do_eval", action="store_true")
    p.add_argument("--per_gpu_train_batch_size", default=16, type=int)
    p.add_argument("--per_gpu_eval_batch_size", default=4, type=int)
    p.add_argument("--gradient_accumulation_steps", default=1, type=int)
    p.add_argument("--learning_rate", default=1e-5, type=float)
    p.add_argument("--weight_decay", default=0.05, type=float, help="parsed, unused (as in the reference :613/:137)")
    p.add_argument("--adam_epsilon", default=1e-5, type=float)
    p.add_argument("--max_grad_norm", default=1.0, type=float)
    p.add_argument("--num_train_epochs", default=30, type=int)
    p.add_argument("--max_steps", default=0, type=int)
    p.add_argument("--logging_steps", default=20, type=int)
    p.add_argument("--valid_steps", default=400, type=int)
    p.add_argument("--epoch_begin", default=1, type=int)
    p.add_argument("--global_step", default=0, type=int)
    p.add_argument("--seed", default=88, type=int)
    p.add_argument("--drop_out", default=0.3, type=float,
                   help="hidden_dropout_prob of both encoders (run_PMR_ModCR.py:585,719,738): live in training mode, also "
                        "inside the frozen encoders; the attention-probability dropout is not applied (DESIGN.md 4.7)")
    p.add_argument("--max_img_seq_length", default=100, type=int)
    p.add_argument("--max_hypo_len", default=80, type=int)
    p.add_argument("--img_feature_dim", default=2054, type=int)
    p.add_argument("--local_rank", default=int(os.environ.get("LOCAL_RANK", 0)), type=int)
    # this build
    p.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    p.add_argument("--roberta_body", default="standin", choices=["standin", "large"],
                   help="large = the 24-layer prefix RoBERTa-large on the HIP kernels, trainable (run_PMR_ModCR.py:772-781; "
                        "random init: local_transformers/roberta-large is not in the reference tree)")
    p.add_argument("--synthetic_train_examples", default=4096, type=int)
    p.add_argument("--synthetic_val_examples", default=256, type=int)
    p.add_argument("--device", default="cuda", type=str)
    return p.parse_args(argv)


def main(argv=None):
    args = get_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(name)s %(message)s")
    modcr_hip.lib()                                    # fail loudly without the HIP library
    if not str(args.device).startswith("cuda") or not torch.cuda.is_available():
        raise SystemExit("run_PMR_ModCR.py needs an MI355X: the ModCR hot path has no CPU fallback "
                         "(the CPU restatement lives under oracle/ and is test infrastructure only)")
    args.world_size = int(os.environ.get("WORLD_SIZE", 1))
    args.rank = int(os.environ.get("RANK", 0))
    args.distributed = args.world_size > 1
    torch.cuda.set_device(args.local_rank)
    args.device = torch.device("cuda", args.local_rank)
    if args.distributed:                               # run_PMR_ModCR.py:438-448 (nccl == RCCL on ROCm)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", init_method="env://")
        dist.barrier()
    os.makedirs(args.output_dir, exist_ok=True)
    torch.manual_seed(args.seed)

    model = tu.build_model(args.device, dtype=args.dtype, seed=args.seed, roberta_body=args.roberta_body,
                           hidden_dropout_prob=args.drop_out)
    import modcr_hip as mh
    mh.DROPOUT.manual_seed(args.seed + 7919 * getattr(args, "rank", 0))   # different masks per rank (different data anyway)
    if args.model_name_or_path:
        sd = torch.load(os.path.join(args.model_name_or_path, "pytorch_model.bin"), map_location="cpu")
        model.calec.global_enc.load_state_dict({k[5:] if k.startswith("bert.") else k: v for k, v in sd.items()}, strict=False)
    if args.seq_model_name_or_path:
        sd = torch.load(args.seq_model_name_or_path, map_location="cpu")
        sd = sd.get("net", sd)
        model.calec.seq_enc.load_state_dict({k[8:]: v for k, v in sd.items() if k.startswith("seq_enc.")}, strict=False)
    if args.eval_model_dir and os.path.isfile(os.path.join(args.eval_model_dir, "model.pth")):
        model.load_state_dict(torch.load(os.path.join(args.eval_model_dir, "model.pth"), map_location="cpu"), strict=False)

    T, R = args.max_hypo_len, args.max_img_seq_length
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


if __name__ == "__main__":
    main()
