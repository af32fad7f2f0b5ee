#!/usr/bin/env python
"""run_vcr_ModCR.py -- VCR Q->A entry point.  The reference's run_vcr_ModCR.py is run_PMR_ModCR.py with
  * the VCR dataset files as defaults (`vcr_data/...`, run_vcr_ModCR.py:487-533),
  * VCR training defaults: batch 8 x 4 accumulation steps (:603-605), validation every 3500 steps (:673), the checkpoint
    tag "VCR-Prefix-tuning_len5_all" (:236) and its --eval_model_dir default,
  * and ONE semantic difference: the RoBERTa body is frozen except its embeddings and pooler (:781-787: every parameter
    whose name contains neither 'embeddings.' nor 'pooler.' gets requires_grad = False) -- so the step's trainable set is
    the heads + roberta.embeddings.* + roberta.pooler.* (~113 M parameters instead of ~416 M).
The model path is identical, so this script re-uses run_PMR_ModCR with those defaults, the freeze as a model hook, and the
VCR-like synthetic shapes of SURVEY 8(d) (T=194 text tokens, R=36 regions, S=230)."""
import sys

import run_PMR_ModCR as pmr

_VCR = "vcr_data/"
VCR_FILE_DEFAULTS = {                                       # run_vcr_ModCR.py:487-533
    "roberta_file_train": _VCR + "vcr_train_CALeC.pkl", "roberta_file_dev": _VCR + "vcr_val_CALeC.pkl",
    "roberta_file_test": _VCR + "vcr_test_CALeC.pkl",
    "clip_file_train": _VCR + "vcr_train.json", "clip_file_dev": _VCR + "vcr_val.json", "clip_file_test": _VCR + "vcr_test.json",
    "vcr_example_file_train": _VCR + "vcr_train_CALeC-o.pkl", "vcr_example_file_dev": _VCR + "vcr_val_CALeC-o.pkl",
    "vcr_example_file_test": _VCR + "vcr_test_CALeC-o.pkl",
    "vcr_feat_file_train": _VCR + "image_feature/train_feat_vcr.pkl", "vcr_feat_file_dev": _VCR + "image_feature/val_feat_vcr.pkl",
    "vcr_feat_file_test": _VCR + "image_feature/test_feat_vcr.pkl",
    "vcr_chunk_mask_train": _VCR + "ChunkMaskTrain_v4_vcr.pkl", "vcr_chunk_mask_dev": _VCR + "ChunkMaskVal_v4_vcr.pkl",
    "vcr_chunk_mask_test": _VCR + "ChunkMaskTest_v4_vcr.pkl",
}


def freeze_roberta_body(model):
    """run_vcr_ModCR.py:781-787 on `model.roberta`: only `embeddings.*` and `pooler.*` stay trainable.  Returns the names
    (relative to the RoBERTa module) that were frozen."""
    frozen = []
    for n, p in model.roberta.named_parameters():
        if 'embeddings.' not in n and 'pooler.' not in n:
            p.requires_grad = False
            frozen.append(n)
    return frozen


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    defaults = ["--per_gpu_train_batch_size", "8", "--gradient_accumulation_steps", "4", "--valid_steps", "3500",
                "--synthetic_text_len", "194", "--synthetic_regions", "36",
                "--eval_model_dir", "output/checkpoint/Tu/VCR-Prefix-tuning_len5_all-3-0.857338351009237-17500.pth"]
    for k, v in VCR_FILE_DEFAULTS.items():
        defaults += ["--" + k, v]
    pmr.CKPT_TAG = "VCR-Prefix-tuning_len5_all"          # run_vcr_ModCR.py:236
    pmr.MODEL_HOOKS = [freeze_roberta_body]
    try:
        return pmr.main(defaults + argv)          # later flags override the defaults
    finally:
        pmr.MODEL_HOOKS = []


if __name__ == "__main__":
    main()
