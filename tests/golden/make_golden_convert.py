"""Golden vector for the LLaVA-OneVision -> HF converter CONTRACT, from the REFERENCE (build container only):
radvlm/evaluation/convert_llava_onevision_weights_to_hf.py -- its KEYS_TO_MODIFY_MAPPING (:49-59) and convert_state_dict_to_hf (:79-90)
are executed from their source lines on the state-dict KEYS this package writes for the Qwen2 + SigLIP flavour (toy geometry).
Writes host_convert_keys.json: {input key -> converted key, dropped keys}."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, HERE)
from make_golden_host import exec_slice, REF  # noqa: E402
from oracle import llava_oracle as O  # noqa: E402
from radvlm_amd.config import GEOMETRIES  # noqa: E402


def main():
    ns = dict(torch=torch)
    path = REF + "/radvlm/evaluation/convert_llava_onevision_weights_to_hf.py"
    exec_slice(path, 49, 59, ns)
    exec_slice(path, 79, 90, ns)
    keys = list(O.param_shapes(GEOMETRIES["toy_qwen"], with_newline=True)) + ["model.layers.0.self_attn.rotary_emb.inv_freq"]
    sd = {k: torch.zeros(1) for k in keys}
    out = ns["convert_state_dict_to_hf"](sd)
    # the conversion is a pure renaming: recover the map by converting one key at a time
    mapping, dropped = {}, []
    for k in keys:
        o = ns["convert_state_dict_to_hf"]({k: torch.zeros(1)})
        if o:
            mapping[k] = next(iter(o))
        else:
            dropped.append(k)
    assert sorted(mapping.values()) == sorted(out)
    assert all(v.dtype == torch.float16 for v in out.values())
    with open(os.path.join(HERE, "host_convert_keys.json"), "w") as f:
        json.dump({"mapping": mapping, "dropped": dropped, "dtype": "float16"}, f, indent=1)
    print(len(mapping), "keys,", len(dropped), "dropped; e.g.", list(mapping.items())[:3])


if __name__ == "__main__":
    main()
