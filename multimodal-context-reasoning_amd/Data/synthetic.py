"""Synthetic PMR / VCR batches that honour the reference's batch contract.

The reference builds its batch dict in `SNLIGPT_gen_collate` (Data/VCRChunkAlign.py:690-741 for
PMR, :903-952 for VCR) from pickles that are not in the tree; the chunk mask / offsets /
gather_index semantics come from utils/GetChunk_v4_vcr.py:107-154.  This module emits the same
keys, shapes and dtypes from a seeded `numpy.random.RandomState` (legacy generator: the stream is
stable across numpy versions) following the recipe in SURVEY.md section 8(d).
"""
import numpy as np
import torch

CLS_ID, SEP_ID = 101, 102
DET_BASE = 30522          # <|det0|>.. tokens appended after the BERT vocab (run_PMR_ModCR.py:709-730)
NUM_DET = 45


def _chunks(rs, n_inner):
    """Partition token positions 1..n_inner into consecutive chunks, lengths ~{1:.5,2:.3,3:.15,4:.05}."""
    chunks, pos = [], 1
    while pos <= n_inner:
        ln = int(rs.choice([1, 2, 3, 4], p=[0.5, 0.3, 0.15, 0.05]))
        ln = min(ln, n_inner - pos + 1)
        chunks.append(list(range(pos, pos + ln)))
        pos += ln
    return chunks


def chunk_mask_from_offsets(mask_len, offsets):
    """utils/GetChunk_v4_vcr.py:107-143: eye + row 0 + last-token row + intra-chunk blocks."""
    m = np.eye(mask_len, dtype=np.float32)
    m[0, :mask_len] = 1
    for ch in offsets:
        for a in ch:
            for b in ch:
                m[a, b] = 1
    m[mask_len - 1, :mask_len] = 1
    return m


def make_batch(num_examples, T=80, R=100, seed=1234, vocab_size=DET_BASE + NUM_DET,
               img_dim=2054, min_text=30, min_regions=20, roberta_len=96, num_choices=4,
               full_length_first=True):
    """Returns the dict `Abstract_Specific.forward(**batch)` consumes (run_PMR_ModCR.py:189-200).

    N = num_examples*num_choices sequences.  With `full_length_first` the first sequence uses the
    whole text/region budget so the padded batch is exactly [N,T] / [N,R] (collate pads to the batch
    maximum, Data/VCRChunkAlign.py:697-713)."""
    rs = np.random.RandomState(seed)
    n = num_examples * num_choices
    input_ids = np.zeros((n, T), np.int64)
    token_type = np.zeros((n, T), np.int64)
    total_label = np.zeros((n, T), np.int64)
    input_mask = np.zeros((n, T + R), np.float32)
    chunk_mask = np.zeros((n, T, T), np.float32)
    img_feat = np.zeros((n, R, img_dim), np.float32)
    label = np.zeros((n,), np.float32)
    gather_index, offsets = [], []
    r_ids = np.ones((n, roberta_len), np.int64)            # RoBERTa pad_token_id = 1
    r_mask = np.zeros((n, roberta_len), np.float32)
    det_hi = min(NUM_DET, max(vocab_size - DET_BASE, 0))
    for e in range(num_examples):
        nreg = R if (e == 0 and full_length_first) else int(rs.randint(min(min_regions, R), R + 1))
        feat = np.maximum(rs.standard_normal((nreg, img_dim - 6)), 0).astype(np.float32)
        box = rs.uniform(0, 1, (nreg, 6)).astype(np.float32)
        answer = int(rs.randint(0, num_choices))
        for c in range(num_choices):
            i = e * num_choices + c
            ln = T if (i == 0 and full_length_first) else int(rs.randint(min(min_text, T), T + 1))
            hi = min(DET_BASE, vocab_size)
            ids = rs.randint(1000 if hi > 1000 else 110, hi, size=ln).astype(np.int64)
            if det_hi > 0:
                det = rs.uniform(size=ln) < 0.05
                det_id = rs.randint(1, det_hi, size=ln)
                ids = np.where(det, DET_BASE + det_id, ids)
                lab = np.where(det, det_id, 0)
            else:
                lab = np.zeros(ln, np.int64)
            mid = ln // 2
            ids[0], ids[mid], ids[ln - 1] = CLS_ID, SEP_ID, SEP_ID
            lab[0] = lab[mid] = lab[ln - 1] = 0
            lab = np.minimum(lab, nreg - 1)          # align target must be a real region column
            input_ids[i, :ln] = ids
            total_label[i, :ln] = lab
            token_type[i, mid + 1:ln] = 1
            input_mask[i, :ln] = 1
            input_mask[i, T:T + nreg] = 1
            img_feat[i, :nreg, :img_dim - 6] = feat
            img_feat[i, :nreg, img_dim - 6:] = box
            ch = _chunks(rs, ln - 2)
            offsets.append(ch)
            gi = np.concatenate([[k] * len(c_) for k, c_ in enumerate(ch)]).astype(np.int64) \
                if ch else np.zeros((0,), np.int64)
            gather_index.append(torch.from_numpy(gi))
            chunk_mask[i, :ln, :ln] = chunk_mask_from_offsets(ln, ch)
            label[i] = 1.0 if c == answer else 0.0
            rl = int(rs.randint(roberta_len // 2, roberta_len + 1)) if i else roberta_len
            r_ids[i, :rl] = rs.randint(4, 50000, size=rl)
            r_ids[i, 0], r_ids[i, rl - 1] = 0, 2
            r_mask[i, :rl] = 1
    align_pos = (total_label != 0).astype(np.int64)
    return {
        "image": None, "text": None,
        "r_input_ids": torch.from_numpy(r_ids),
        "r_token_type_ids": torch.zeros(n, roberta_len, dtype=torch.int64),
        "r_attention_mask": torch.from_numpy(r_mask),
        "input_ids": torch.from_numpy(input_ids),
        "token_type_ids": torch.from_numpy(token_type),
        "input_mask": torch.from_numpy(input_mask),
        "img_feat": torch.from_numpy(img_feat),
        "label": torch.from_numpy(label),
        "chunk_attention_mask": torch.from_numpy(chunk_mask),
        "gather_index": gather_index,
        "offsets": offsets,
        "total_label": torch.from_numpy(total_label),
        "align_pos": torch.from_numpy(align_pos),
    }


class SyntheticPMRDataset(torch.utils.data.Dataset):
    """Stands in for PMR_ChunkAlign_Dataset_align_ensemble_T (Data/VCRChunkAlign.py:529-688): one
    item = one example (4 choices); `collate` = SNLIGPT_gen_collate's output contract."""

    def __init__(self, num_examples, T=80, R=100, seed=1234, **kw):
        self.num_examples, self.T, self.R, self.seed, self.kw = num_examples, T, R, seed, kw

    def __len__(self):
        return self.num_examples

    def __getitem__(self, i):
        return i

    def SNLIGPT_gen_collate(self, idx):
        return make_batch(len(idx), self.T, self.R, seed=self.seed + 7919 * int(idx[0]), **self.kw)
