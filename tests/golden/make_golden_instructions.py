"""Build-container script (needs /root/reference): pins radvlm_amd/data/create_instructions.py to the reference's templated instruction
generators (radvlm/data/create_instructions.py:9-26 helpers, :120-529 generators).

Two outputs, both DATA:
  radvlm_amd/data/instruction_templates.json   the prompt vocabulary: the question / answer / prefix string lists of every generator, read
                                               out of the reference module's syntax tree (the strings are the training-data contract: a model
                                               fine-tuned on RadVLM data has seen exactly these prompts, in this order under `random.choice`)
  tests/golden/instruction_generators.json     inputs and outputs of the reference's own functions, imported in place and run under
                                               seeded `random` (tests/test_host_golden.py replays them through the restatement)
The reference module is imported from the read-only tree with its two heavy imports stubbed (radvlm/__init__ needs openai + DATA_DIR,
radvlm.data.utils needs torchvision); nothing of it is copied.
"""
import ast
import importlib
import json
import os
import random
import sys
import types

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def load_reference_module():
    sys.dont_write_bytecode = True
    for name, path in (("radvlm", REF + "/radvlm"), ("radvlm.data", REF + "/radvlm/data")):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
    u = types.ModuleType("radvlm.data.utils")
    u.custom_collate_fn = lambda batch: batch
    sys.modules["radvlm.data.utils"] = u
    return importlib.import_module("radvlm.data.create_instructions")


def template_tables(path):
    """{function: {variable: [strings]}} for every list-of-string-literals assignment inside the generator functions."""
    tree = ast.parse(open(path).read())
    out = {}
    for fn in tree.body:
        if not isinstance(fn, ast.FunctionDef) or not fn.name.startswith("generate_instruction_"):
            continue
        tabs = {}
        for node in ast.walk(fn):
            if isinstance(node, ast.Assign) and isinstance(node.value, ast.List) and node.value.elts and \
                    all(isinstance(e, ast.Constant) and isinstance(e.value, str) for e in node.value.elts):
                name = node.targets[0].id
                # `answer_variations` is assigned twice in generate_instruction_abnormalities (several / one abnormality)
                key = name if name not in tabs else name + "_single"
                tabs[key] = [e.value for e in node.value.elts]
        out[fn.name[len("generate_instruction_"):]] = tabs
    return out


def main():
    ref = load_reference_module()
    tabs = template_tables(REF + "/radvlm/data/create_instructions.py")
    with open(os.path.join(ROOT, "radvlm_amd", "data", "instruction_templates.json"), "w") as f:
        json.dump(tabs, f, indent=1, ensure_ascii=False)
    boxes = [[0.1234, 0.25, 0.5, 0.75], [0.3333333, 0.05, 0.61, 0.4449], [0.0, 0.0, 1.0, 1.0]]
    cases = {
        "format_boxes": [dict(args=[boxes[:n]], kwargs=kw) for n in (1, 2, 3) for kw in ({}, {"num_float": 3})],
        "select_article": [dict(args=[w], kwargs={}) for w in ("Atelectasis", "effusion", "Opacity", "nodule", "Umbrella")],
        "generate_instruction_report_generation": [dict(args=["Heart size is normal. No focal consolidation."], kwargs=kw)
                                                   for kw in ({}, {"german_suffixe": True})],
        "generate_instruction_phrase_location": [dict(args=[boxes[:n], lab], kwargs={})
                                                 for n, lab in ((1, "Small left pleural effusion"), (2, "ET tube in place"), (1, "CABG"))],
        "generate_instruction_location": [dict(args=[boxes[:n], lab], kwargs={})
                                          for n, lab in ((1, "Right lung"), (2, "cardiac silhouette"), (1, "SVC"))],
        "generate_instruction_abnormalities_grouped": [dict(args=[boxes[:len(ab)], ab], kwargs={})
                                                       for ab in ([], ["Atelectasis"], ["Nodule", "Effusion", "Nodule"])],
        "generate_instruction_foreign_objects": [dict(args=[boxes[:n]], kwargs={}) for n in (0, 1, 3)],
        "generate_instruction_abnormalities": [dict(args=[ab], kwargs={})
                                               for ab in ([], ["Cardiomegaly"], ["Edema", "Effusion", "Edema", "Atelectasis"])],
    }
    out = {}
    for fn, calls in cases.items():
        recs = []
        for c in calls:
            for seed in range(12):
                random.seed(seed)
                res = getattr(ref, fn)(*c["args"], **c["kwargs"])
                # what `random` hands out next tells whether the restatement consumed the stream identically
                recs.append(dict(args=c["args"], kwargs=c["kwargs"], seed=seed, result=res, next_random=random.random()))
        out[fn] = recs
    # the error the reference raises on mismatched lengths
    try:
        ref.generate_instruction_abnormalities_grouped(boxes[:1], ["a", "b"])
    except ValueError as e:
        out["grouped_length_mismatch_error"] = str(e)
    with open(os.path.join(HERE, "instruction_generators.json"), "w") as f:
        json.dump(out, f, indent=1, ensure_ascii=False)
    print({k: (len(v) if isinstance(v, list) else v) for k, v in out.items()}, {k: {n: len(t) for n, t in v.items()} for k, v in tabs.items()})


if __name__ == "__main__":
    main()
