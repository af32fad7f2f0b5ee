"""Host-side batch packing for the ModCR hot path: the batch contract of the reference's
`SNLIGPT_gen_collate` (Data/VCRChunkAlign.py:690-741) built ONCE per batch on the host, in pinned memory, and moved
with a handful of asynchronous copies -- instead of the reference's per-sample `.cuda()` tensors (`:596-681`: ~20 tiny
device tensors per choice, 4 choices per example) padded and stacked on the device.

Input = what the reference's datasets return: a list of examples, each a tuple of per-choice 19-tuples
    (img_id, image, text, r_input_ids, r_segment_ids, r_input_mask, input_ids, segment_ids, input_mask, img_feat,
     img_mask, target, chunk_mask, gather_index, offsets, ques, ans, total_label, align_pos)
with CPU tensors (or anything `torch.as_tensor` takes).  Output = the same dict (same keys, shapes, dtypes, padding
values; regions truncated to the batch maximum as `:712-716`), with two host-packed additions the kernels read directly:
    gather_index  -> int32 [N, T] chunk-id rows (-1 = leave the query alone) instead of a list of N ragged tensors
                     (the list form is kept under 'gather_index_list' for callers that want the reference's type)
    label         -> float32 [N] (the reference's `target.type(FloatTensor)`)
`to_device(batch, device)` issues one non-blocking copy per tensor from the pinned staging buffers.
"""
import numpy as np
import torch

FIELDS = ("img_id", "image", "text", "r_input_ids", "r_segment_ids", "r_input_mask", "input_ids", "segment_ids",
          "input_mask", "img_feat", "img_mask", "target", "chunk_mask", "gather_index", "offsets", "ques", "ans",
          "total_label", "align_pos")


def _pin(t):
    if torch.cuda.is_available():
        try:
            return t.pin_memory()
        except RuntimeError:
            return t
    return t


def _pad_stack(seqs, dtype, pad=0):
    """pad_sequence(batch_first=True) into one pinned buffer"""
    seqs = [torch.as_tensor(s).reshape(-1) for s in seqs]
    width = max(int(s.numel()) for s in seqs)
    out = torch.full((len(seqs), width), pad, dtype=dtype)
    for i, s in enumerate(seqs):
        out[i, :s.numel()] = s.to(dtype)
    return _pin(out)


def SNLIGPT_gen_collate(inputs, pack=True):
    choices = [c for example in inputs for c in example]            # unzip(concat(inputs)) of the reference
    col = {k: [c[i] for c in choices] for i, k in enumerate(FIELDS)}
    n = len(choices)
    r_input_ids = _pad_stack(col["r_input_ids"], torch.int64)
    r_input_mask = _pad_stack(col["r_input_mask"], torch.float32)
    r_segment_ids = _pad_stack(col["r_segment_ids"], torch.int64)
    input_ids = _pad_stack(col["input_ids"], torch.int64)
    segment_ids = _pad_stack(col["segment_ids"], torch.int64)
    text_mask = _pad_stack(col["input_mask"], torch.float32)
    total_label = _pad_stack(col["total_label"], torch.int64)
    align_pos = _pad_stack(col["align_pos"], torch.int64)
    target = _pin(torch.as_tensor(np.asarray([float(torch.as_tensor(t)) for t in col["target"]], np.float32)))
    img_mask = torch.stack([torch.as_tensor(m).to(torch.float32) for m in col["img_mask"]], 0)
    max_img = int(img_mask.sum(-1).max().item())                       # :714 batch-maximum region count
    img_mask = img_mask[:, :max_img]
    img_feat = _pin(torch.stack([torch.as_tensor(f).to(torch.float32)[:max_img] for f in col["img_feat"]], 0).contiguous())
    input_mask = _pin(torch.cat((text_mask, img_mask), -1).contiguous())
    t = input_ids.shape[1]
    chunk = torch.zeros((n, t, t), dtype=torch.float32)
    for i, m in enumerate(col["chunk_mask"]):                            # zero-padded to [max_hypo, max_hypo] (:718-725)
        m = torch.as_tensor(m).to(torch.float32)
        chunk[i, :m.shape[0], :m.shape[1]] = m
    chunk = _pin(chunk)
    gi_list = [torch.as_tensor(g).to(torch.int64).reshape(-1) for g in col["gather_index"]]
    batch = {"img_id": col["img_id"], "image": None, "text": None,
             "r_input_ids": r_input_ids, "r_token_type_ids": r_segment_ids, "r_attention_mask": r_input_mask,
             "input_ids": input_ids, "token_type_ids": segment_ids, "input_mask": input_mask, "img_feat": img_feat,
             "label": target, "ques_str": col["ques"], "ans_str": col["ans"], "chunk_attention_mask": chunk,
             "gather_index": gi_list, "offsets": col["offsets"], "total_label": total_label, "align_pos": align_pos}
    if pack:
        cid = torch.full((n, t), -1, dtype=torch.int32)
        for i, g in enumerate(gi_list):
            k = min(int(g.numel()), t - 1)
            cid[i, 1:1 + k] = g[:k].to(torch.int32)                      # text token j (1-based, after [CLS]) -> chunk g[j-1]
        batch["gather_index_list"] = gi_list
        batch["gather_index"] = _pin(cid)
    return batch


def to_device(batch, device):
    out = {}
    for k, v in batch.items():
        if torch.is_tensor(v):
            out[k] = v.to(device, non_blocking=True)
        elif isinstance(v, list) and v and torch.is_tensor(v[0]) and k != "gather_index_list":
            out[k] = [t.to(device, non_blocking=True) for t in v]
        else:
            out[k] = v
    return out
