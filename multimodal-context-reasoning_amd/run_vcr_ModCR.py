#!/usr/bin/env python
"""run_vcr_ModCR.py -- VCR Q->A entry point.  The reference's run_vcr_ModCR.py is run_PMR_ModCR.py
with the VCR dataset class, VCR defaults (batch 8 x 4 accumulation steps, validation every 3500
steps, :603-605,:673) and the checkpoint tag "VCR-Prefix-tuning_len5_all" (:236); RoBERTa is frozen
there except embeddings / pooler (:781-787), which changes only which parameters get gradients.
The model path is identical, so this script re-uses run_PMR_ModCR with those defaults and the
VCR-like synthetic shapes of SURVEY 8(d) (T=194 text tokens, R=36 regions, S=230)."""
import sys

import run_PMR_ModCR as pmr


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    defaults = ["--per_gpu_train_batch_size", "8", "--gradient_accumulation_steps", "4", "--valid_steps", "3500",
                "--synthetic_text_len", "194", "--synthetic_regions", "36",
                "--eval_model_dir", "output/checkpoint/Tu/VCR-Prefix-tuning_len5_all-3-0.857338351009237-17500.pth"]
    pmr.CKPT_TAG = "VCR-Prefix-tuning_len5_all"          # run_vcr_ModCR.py:236
    return pmr.main(defaults + argv)          # later flags override the defaults


if __name__ == "__main__":
    main()
