#!/usr/bin/env python
"""bench.py with module-level A/B flags of modeling/hip_layers.py (or modeling/<module>.<FLAG>) changed first (tools only; the product reads no environment):
    FLAGS="PACK_FUSED=0,KEEP_GELU_INPUT=1" python tools/ab_flags.py --train-encoders --steps 20 --no-extra-legs ..."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "multimodal-context-reasoning_amd"))
import importlib  # noqa: E402

for kv in filter(None, os.environ.get("FLAGS", "").split(",")):
    k, v = kv.split("=")
    modname, _, attr = k.rpartition(".")                # "PACK_FUSED" (hip_layers) or "modeling_ensemble.BATCH_GLOBAL_PASSES"
    mod = importlib.import_module("modeling." + (modname or "hip_layers"))
    if not hasattr(mod, attr):
        raise SystemExit("no such flag: %s" % k)
    setattr(mod, attr, type(getattr(mod, attr))(int(v)))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(sys.argv[0], run_name="__main__")
