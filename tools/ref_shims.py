"""Import the ModCR reference (/root/reference) in THIS container so golden vectors can be generated.

Test-infrastructure only.  Nothing under tools/ is imported by the product, by tests that run on
the GPU box, by bench.py or by smoke(): /root/reference does not exist there.  The shim list
follows SURVEY.md section 8(c): the reference was written against transformers ~4.6 and a missing
`local_transformers` package, so a handful of names have to be put back before its modules import.
"""
import os
import sys
import tempfile
import types
import zipfile

REFERENCE_ROOT = os.environ.get("MODCR_REFERENCE", "/root/reference")


def _install_shims():
    import torch
    import transformers
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu

    for name in ("apply_chunking_to_forward", "prune_linear_layer", "Conv1D"):
        if not hasattr(mu, name):
            setattr(mu, name, getattr(pu, name))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        mu.find_pruneable_heads_and_indices = getattr(
            pu, "find_pruneable_heads_and_indices", lambda *a, **k: (set(), None))
    for name in ("prune_conv1d_layer", "SequenceSummary"):
        if not hasattr(mu, name):
            setattr(mu, name, type(name, (), {}))
    for name, val in (("WEIGHTS_NAME", "pytorch_model.bin"), ("TF_WEIGHTS_NAME", "model.ckpt")):
        if not hasattr(mu, name):
            setattr(mu, name, val)
        if not hasattr(transformers, name):
            setattr(transformers, name, val)
    if not hasattr(transformers, "BeamScorer"):
        transformers.BeamScorer = type("BeamScorer", (), {})
    import transformers.file_utils as fu
    if not hasattr(fu, "cached_path"):
        fu.cached_path = lambda *a, **k: None

    import transformers.generation as gen
    for modname in ("transformers.generation_logits_process",
                    "transformers.generation_stopping_criteria",
                    "transformers.generation_beam_search"):
        if modname not in sys.modules:
            m = types.ModuleType(modname)

            def _getattr(name, _gen=gen):
                try:
                    return getattr(_gen, name)
                except AttributeError:
                    return type(name, (), {})
            m.__getattr__ = _getattr
            sys.modules[modname] = m

    if "anytree" not in sys.modules:
        m = types.ModuleType("anytree")
        m.AnyNode = type("AnyNode", (), {})
        m.__getattr__ = lambda name: type(name, (), {})
        sys.modules["anytree"] = m

    pkg = "local_transformers"
    chain = [pkg, pkg + ".adapter_transformers", pkg + ".adapter_transformers.models",
             pkg + ".adapter_transformers.models.roberta"]
    for name in chain:
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            sys.modules[name] = m
    rob = sys.modules[chain[-1]]
    for name in ("RobertaModel", "RobertaConfig", "RobertaTokenizer"):
        setattr(rob, name, type(name, (), {}))

    # old-style init: the reference calls self.init_weights() (modeling_transfomres.py:598, v10:257)
    transformers.PreTrainedModel.init_weights = lambda self: self.apply(self._init_weights)
    # no GPU here: v10:71,74 call .cuda(device) on freshly built tensors
    torch.Tensor.cuda = lambda self, *a, **k: self


def load_reference_collate():
    """The reference's PMR dataset class (Data/VCRChunkAlign.py:529-741) for its SNLIGPT_gen_collate.  Its module imports
    toolz.sandbox.unzip / cytoolz.concat (absent here: given their published semantics -- unzip = inverse of zip, concat =
    itertools.chain.from_iterable), `clip`, and through Data/data.py lmdb / lz4 / msgpack_numpy (readers this path never
    calls: empty placeholders).  `.cuda(...)` is the identity (load_reference's shim), so the collate runs on the CPU."""
    import importlib
    import itertools
    load_reference()
    def unzip(seq):
        seq = list(seq)
        return tuple(iter(col) for col in zip(*seq)) if seq else ()
    for name, attrs in (("toolz", {}), ("toolz.sandbox", {"unzip": unzip}), ("cytoolz", {"concat": itertools.chain.from_iterable}),
                        ("clip", {}), ("lmdb", {}), ("lz4", {}), ("lz4.frame", {"compress": None, "decompress": None}),
                        ("msgpack_numpy", {"patch": lambda: None})):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = []
            for k, v in attrs.items():
                setattr(m, k, v)
            sys.modules[name] = m
    for k in [k for k in sys.modules if k == "Data" or k.startswith("Data.")]:      # the repo's own Data package must not win
        del sys.modules[k]
    sys.path[:] = [q for q in sys.path if not os.path.exists(os.path.join(q or ".", "Data", "synthetic.py"))]
    mod = importlib.import_module("Data.VCRChunkAlign")
    assert mod.__file__.startswith(REFERENCE_ROOT), mod.__file__
    return mod


_LOADED = {}


def load_reference():
    """Returns a namespace with the reference's hot-path modules imported."""
    if _LOADED:
        return _LOADED["ns"]
    _install_shims()
    tmp = tempfile.mkdtemp(prefix="modcr_atf_")
    with zipfile.ZipFile(os.path.join(REFERENCE_ROOT, "a_transformers.zip")) as z:
        z.extractall(tmp)
    # the reference's `modeling/` has no __init__.py (namespace package): a regular package of the
    # same name anywhere on sys.path (this repo's drop-in tree) would win regardless of order
    sys.path[:] = [q for q in sys.path if not os.path.exists(os.path.join(q or ".", "modeling", "__init__.py"))]
    sys.path.insert(0, tmp)
    sys.path.insert(0, REFERENCE_ROOT)
    # the repo's own drop-in tree also has a top-level `modeling` package: make sure the
    # reference's wins inside this process
    for k in [k for k in sys.modules if k == "modeling" or k.startswith("modeling.")]:
        del sys.modules[k]
    import importlib
    ns = types.SimpleNamespace()
    ns.a_bert = importlib.import_module("a_transformers.modeling_bert")
    ns.m_bert = importlib.import_module("modeling.modeling_bert")
    ns.m_tr = importlib.import_module("modeling.modeling_transfomres")
    ns.v10 = importlib.import_module("modeling.modeling_vcr_chunkalign_v10")
    ns.ens = importlib.import_module("modeling.modeling_ensemble")
    assert ns.v10.__file__.startswith(REFERENCE_ROOT), ns.v10.__file__
    _LOADED["ns"] = ns
    return ns


def make_ref_config(ns, **kw):
    """BertConfig with every attribute the reference reads (run_PMR_ModCR.py:717-748, v10:158-169)."""
    cfg = ns.a_bert.BertConfig()
    defaults = dict(
        vocab_size=30522 + 45, hidden_size=768, num_hidden_layers=12, num_attention_heads=12,
        intermediate_size=3072, hidden_act="gelu", hidden_dropout_prob=0.0,
        attention_probs_dropout_prob=0.0, max_position_embeddings=512, type_vocab_size=2,
        initializer_range=0.02, layer_norm_eps=1e-12, pad_token_id=0,
        position_embedding_type="absolute", is_decoder=False, add_cross_attention=False,
        chunk_size_feed_forward=0, gradient_checkpointing=False,
        img_feature_dim=2054, img_feature_type="frcnn", use_img_layernorm=1,
        img_layer_norm_eps=1e-12, output_attentions=True, output_hidden_states=False,
        max_hypo=50, add_residual=False, add_local_residual=False, use_cache=False)
    defaults.update(kw)
    for k, v in defaults.items():
        setattr(cfg, k, v)
    return cfg
