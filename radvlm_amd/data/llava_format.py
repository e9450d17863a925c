"""Flatten instruction datasets into the LLaVA JSON the trainer reads.

Restates reference radvlm/data/create_instructions.py:29-71 (create_json_cell_llava) and :75-116
(generate_llava_dataset_from_instruction_dataset) and the repetition-factor mixing of
radvlm/data/create_llava_dataset.py:219-332.  Record contract:
    {"image": <path>, "conversations": [{"from": "human"|"gpt", "value": str}, ...], "id": str}
with "<image>\\n" prefixed to the first human turn.  Works on any dataset object yielding the reference's sample
dicts ({img_path, instr: {question, answer} | conversation: [...]}); the CXR dataset classes themselves need the
(absent, licensed) image corpora and stay outside this build.
"""
import json
import random

import numpy as np


def create_json_cell_llava(sample, id_prefix, sample_idx, dataset):
    """One LLaVA record. `sample` carries either "conversation" (list of {from,value} turns or of
    {question,answer} pairs) or "instr" ({question,answer}); the FIRST turn gets the "<image>\n" prefix."""
    turns = sample["conversation"] if "conversation" in sample else sample["instr"]
    if isinstance(turns, dict):
        turns = [turns]
    conv = []
    for j, t in enumerate(turns):
        lead = "<image>\n" if j == 0 else ""
        if "from" in t and "value" in t:
            conv.append({**t, "value": lead + t["value"]})
        elif "question" in t and "answer" in t:
            conv.append({"from": "human", "value": lead + t["question"]})
            conv.append({"from": "gpt", "value": t["answer"]})
    cell = {"image": sample["img_path"], "conversations": conv, "id": f"{id_prefix}_{sample_idx}"}
    if "labels" in sample:
        cell["labels"] = sample["labels"]
    if getattr(dataset, "pathologies", None):
        cell["pathologies"] = list(dataset.pathologies)
    return cell


def generate_llava_dataset_from_instruction_dataset(dataset_info, batch_size=64, num_workers=0, seed=0):
    """dataset_info: list of {dataset, id_prefix, num_samples?}.  Each dataset is visited in a shuffled order and
    contributes up to num_samples records; ids run over the whole output (`<prefix>_<global index>`), and
    numpy/random are reseeded per dataset like the reference (the reference's own visiting order comes from an
    unseeded shuffling DataLoader, so only the record format is a contract, not the order)."""
    out = []
    for k, info in enumerate(dataset_info):
        np.random.seed(seed)
        random.seed(seed)
        ds = info["dataset"]
        prefix = info.get("id_prefix", k)
        limit = info.get("num_samples", len(ds))
        order = list(range(len(ds)))
        random.Random(seed + k).shuffle(order)
        taken = 0
        for i in order:
            if taken >= limit:
                break
            sample = ds[i]
            if sample is None:   # the reference's collate drops failed samples
                continue
            out.append(create_json_cell_llava(sample, prefix, len(out), ds))
            taken += 1
    return out


def mix_datasets(dataset_info_with_repeats):
    """Expand (info, repetition) pairs: a repetition factor r contributes r copies of the dataset's cells."""
    mixed = []
    for info, rep in dataset_info_with_repeats:
        cells = generate_llava_dataset_from_instruction_dataset([info])
        for _ in range(int(rep)):
            mixed.extend(cells)
    return mixed


def write_llava_json(cells, path):
    with open(path, "w") as f:
        json.dump(cells, f, indent=1)
