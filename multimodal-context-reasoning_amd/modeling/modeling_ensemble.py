"""Drop-in for Abstract_Specific (reference modeling/modeling_ensemble.py:424-539): frozen image-only
global_enc pass, the two mapping networks, ChunkAlign ensemble, prefix-conditioned RoBERTa (caller
supplied), Linear(1024,1) scorer, view(-1,4), soft-label cross entropy.  Same constructor and
forward signature, same return tuple (loss, (None, None, loss_abstract, None), logits[B,4])."""
import torch
from torch import nn

import modcr_hip as mh
from . import hip_autograd as ag


BATCH_GLOBAL_PASSES = True       # default of Abstract_Specific.batch_global_passes (tools may clear it for an A/B run)


class _MappingNetwork(nn.Sequential):
    """Dropout -> Linear(768,3840) -> Tanh -> Dropout -> Linear(3840,5120)  (modeling_ensemble.py:439-457);
    indices 1 and 4 carry the parameters, as in the reference's nn.Sequential."""

    def __init__(self, width=768):
        # width: the reference hard-codes 768 (Oscar-base); the encoder width of the model otherwise (BASELINE configs[4])
        super().__init__(nn.Dropout(p=0.1), nn.Linear(width, width * 5, bias=True), nn.Tanh(), nn.Dropout(p=0.1),
                         nn.Linear(width * 5, 1024 * 5, bias=True))

    bf16 = True      # MFMA GEMMs on bf16 copies of the fp32 parameters; False = exact-fp32 parity path

    def forward(self, x):
        # fp32 activations -> (bf16 mode) 3-term bf16 split inside ag.linear: the prefixes feed the answer logits
        # directly, so the mappers are kept fp32-accurate on the MFMA path (M = N sequences: tiny).
        # Dropout(0.1) in front of both Linear layers in training mode (modeling_ensemble.py:440,443).
        x = ag.dropout(x, self[0].p, self.training)
        h = ag.linear(x, self[1].weight, self[1].bias, act=mh.ACT_TANH)
        h = ag.dropout(h, self[3].p, self.training)
        return ag.linear(h, self[4].weight, self[4].bias)


class Abstract_Specific(nn.Module):
    def __init__(self, roberta_model, calec_model, clip_model=None, num_labels=4):
        super(Abstract_Specific, self).__init__()
        self.num_labels = num_labels
        self.calec = calec_model
        self.roberta = roberta_model
        w = getattr(getattr(calec_model.global_enc, "config", None), "hidden_size", 768)     # 768 in the reference (modeling_ensemble.py:433-453)
        if clip_model is not None:
            self.clip_model = clip_model
            self.classifier = nn.Linear(1024 + w + 512, 1)
        else:
            self.classifier = nn.Linear(w + w, 1)
        self.abst_confidence_scorer = nn.Linear(1024, 1)
        self.confidence_scorer = nn.Linear(w, 1)
        self.mapping_network_alignment = _MappingNetwork(w)
        self.mapping_network_vision = _MappingNetwork(w)
        self.promptfuse = torch.nn.Embedding(2, 1024)
        # batch_global_passes: this module's image-only global_enc pass and calec's full pass -- the SAME frozen encoder -- run as one
        # batch of rows through the token-wise blocks (BertImgModel.forward_pair; attention per pass).  143 872 rows fill the GEMM
        # rounds better than 92 160 + 51 712 (the N = 768 products of the image-only pass alone end in a third-full round).  Round 1
        # measured it slower (34.2 vs 33.5 ms at 64 examples, that round's kernels: fp32 pre-LayerNorm rows, no column-group walk);
        # with the round-6 kernels it is faster in every one of three interleaved rounds, 48.69 / 48.69 / 48.64 -> 48.41 / 48.39 /
        # 48.39 ms per step (profiles/r06_ab_batch_global_passes.log), and it is the default.  Eval-mode outputs are those of the two
        # separate calls bit for bit (every row-wise kernel computes a row independently of its neighbours:
        # tests/test_hip_models.py::test_batched_global_enc_passes_equal_separate_passes); in training mode the dropout counters are
        # handed out in a different order (one range over all rows per sublayer instead of one per pass): other masks, same law.
        # Not taken with trainable encoders (the full pass needs its graph), with config.modcr_last_layer_rows, or when the
        # image-only sequences are short enough to be packed several to an attention tile (VCR: 1 + 36 rows).
        self.batch_global_passes = BATCH_GLOBAL_PASSES        # tools / tests set it to exercise BertImgModel.forward_pair
        fp32 = getattr(getattr(calec_model.global_enc, "config", None), "modcr_dtype", "bf16") == "fp32"
        self.mapping_network_alignment.bf16 = self.mapping_network_vision.bf16 = not fp32

    def forward(self, image, text, roberta_input_ids, roberta_token_type_ids, roberta_attention_mask, input_ids,
                img_feat, input_mask=None, token_type_ids=None, position_ids=None, head_mask=None,
                encoder_history_states=None, offsets=None, chunk_attention_mask=None, gather_index=None,
                label=None, align_pos=None, total_label=None):
        n = input_ids.size(0)
        ag.set_exact(not self.mapping_network_vision.bf16)
        from .modeling_transfomres import ImgEmbedMixin, PACK_SHORT
        ImgEmbedMixin._epoch += 1      # the region-embedding re-use of the three encoder passes below never spans two calls
        # vision representations (modeling_ensemble.py:466-475)
        global_outputs = None
        grad_outside = torch.is_grad_enabled()
        with torch.no_grad():
            img_attention_mask = torch.cat([input_mask[:, :1], input_mask[:, -img_feat.size(1):]], dim=-1)
            genc = self.calec.global_enc
            pair = (getattr(genc, "forward_pair", None) if self.batch_global_passes and head_mask is None
                    and encoder_history_states is None and position_ids is None and input_mask is not None
                    and not (getattr(genc, "trainable", False) and grad_outside)
                    and not getattr(getattr(genc, "config", None), "modcr_last_layer_rows", False)
                    and 1 + img_feat.size(1) > PACK_SHORT else None)     # (shorter image-only sequences are packed several to a tile)
            if pair is not None:
                # this image-only pass and calec's full pass run the same frozen encoder: one batch of rows through the
                # token-wise blocks, attention per pass (BertImgModel.forward_pair)
                global_outputs, image_features_ = pair(input_ids, token_type_ids, input_mask, img_feat, img_attention_mask)
            else:
                # (opt-in, config.modcr_last_layer_rows: only the [CLS] row of this pass is read, two lines below)
                lr = ({"modcr_last_rows": 1} if getattr(getattr(self.calec.global_enc, "config", None), "modcr_last_layer_rows", False)
                      and not (getattr(self.calec.global_enc, "trainable", False) and torch.is_grad_enabled()) else {})
                image_features_ = self.calec.global_enc(input_ids[:, :1], img_feats=img_feat,
                                                        attention_mask=img_attention_mask, position_ids=None,
                                                        token_type_ids=None, head_mask=None, encoder_history_states=None, **lr)
            img_cls = mh.convert(image_features_[0][:, 0, :], mh.F32)
        prefix_vision = self.mapping_network_vision(img_cls).reshape(n, 5, 1024)
        vision_mask = input_mask[:, :1].repeat(1, 5)

        CALeC_encoder_output, align_loss, specific_alignment = self.calec(
            input_ids=input_ids, img_feat=img_feat, input_mask=input_mask, token_type_ids=token_type_ids,
            position_ids=position_ids, head_mask=head_mask, encoder_history_states=encoder_history_states,
            offsets=offsets, chunk_attention_mask=chunk_attention_mask, gather_index=gather_index,
            align_pos=align_pos, total_label=total_label, abstract_hidden_states=None,
            **({"global_outputs": global_outputs} if global_outputs is not None else {}))

        Alignment_prompt = self.mapping_network_alignment(CALeC_encoder_output).unsqueeze(1).view(n, 5, 1024)
        align_mask = input_mask[:, :1].repeat(1, 5)
        prefix_emb = torch.cat([prefix_vision, Alignment_prompt], dim=1)
        prompt_mask = torch.cat([vision_mask, align_mask], dim=1)

        roberta_encoder_outputs = self.roberta(input_ids=roberta_input_ids, token_type_ids=roberta_token_type_ids,
                                               attention_mask=roberta_attention_mask, prompt_embeddings=prefix_emb,
                                               input_mask=prompt_mask)
        abstract_level = roberta_encoder_outputs[1]
        abst_logit = ag.linear(abstract_level, self.abst_confidence_scorer.weight, self.abst_confidence_scorer.bias)
        reshaped_logits = abst_logit.view(-1, self.num_labels)
        loss = None
        loss_specific = None
        loss_abstract = None
        align_f_loss = None
        if label is not None:
            label = label.view(reshaped_logits.size())
            loss = ag.McCeFn.apply(reshaped_logits, label)
            loss_abstract = loss          # the reference evaluates the same CE twice (modeling_ensemble.py:536-537)
        return loss, (None, loss_specific, loss_abstract, align_f_loss), reshaped_logits
