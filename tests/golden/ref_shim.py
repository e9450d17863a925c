"""Loader for the reference's hot-path modules (build container only; needs /root/reference).

Test infrastructure: this is how golden vectors were produced (SURVEY.md section 8c recipe).  It registers
synthetic parent packages so the reference's package ``__init__`` files (which pull in absent
third-party packages) are skipped, and aliases three helpers that moved between transformers 4.40
and 5.x.  ``llava.conversation`` is never imported (it performs a hub lookup at import time).
Nothing from the reference is copied: modules are imported in place from the read-only tree.
"""
import importlib
import importlib.util
import sys
import types

REF = "/root/reference/finetuning"


def load_reference():
    sys.dont_write_bytecode = True
    import transformers.modeling_utils as mu
    import transformers.pytorch_utils as pu
    for n in ("apply_chunking_to_forward", "prune_linear_layer"):
        if not hasattr(mu, n):
            setattr(mu, n, getattr(pu, n))
    if not hasattr(mu, "find_pruneable_heads_and_indices"):
        mu.find_pruneable_heads_and_indices = lambda *a, **k: (set(), None)
    for name, path in (("llava", REF + "/llava"), ("llava.model", REF + "/llava/model"),
                       ("llava.model.language_model", REF + "/llava/model/language_model")):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__path__ = [path]
            sys.modules[name] = m
    mods = {}
    for n in ("llava.constants", "llava.utils", "llava.mm_utils", "llava.model.llava_arch",
              "llava.model.language_model.llava_llama"):
        mods[n.split(".")[-1]] = importlib.import_module(n)
    return mods


def load_vendored_llama():
    """The in-tree (dead-by-import) modeling_llama.py, loaded standalone by file path."""
    p = REF + "/llava/model/language_model/modeling_llama.py"
    spec = importlib.util.spec_from_file_location("ref_modeling_llama", p)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m
