"""Templated single-turn instructions for the CXR tasks (report generation, phrase / region grounding, abnormality detection and
classification, foreign objects) -- the corpus-free half of the reference's instruction-dataset generators.

Restates reference radvlm/data/create_instructions.py:9-26 (box / article helpers) and :120-529 (the six `generate_instruction_*`
functions the dataset classes of radvlm/data/datasets.py call per sample).  The prompt vocabulary lives in
``instruction_templates.json`` (data: the strings a RadVLM-tuned model has seen); the logic here is table-driven.  Contract kept:
  * every function returns {"question": str, "answer": str};
  * under a seeded ``random`` the same text comes out as from the reference, and the generator is left in the same state (each
    function draws from ``random.choice`` in the reference's order) -- pinned by tests/golden/instruction_generators.json, which was
    produced by the reference's own functions (tests/golden/make_golden_instructions.py).
"""
import json
import os
import random
from collections import OrderedDict

from .llava_format import create_json_cell_llava, generate_llava_dataset_from_instruction_dataset  # noqa: F401  (same module in the reference)

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "instruction_templates.json"), encoding="utf-8") as _f:
    TEMPLATES = json.load(_f)


def format_boxes(bounding_boxes, num_float=2):
    """'[x1, y1, x2, y2]' per box, rounded to num_float digits, joined 'a, b and c' (reference :9-19).  An empty list is an error there
    too (it indexes the last element)."""
    items = ["[" + ", ".join(str(round(v, num_float)) for v in box[:4]) + "]" for box in bounding_boxes]
    if len(items) < 2:
        return items[-1]
    return ", ".join(items[:-1]) + " and " + items[-1]


def select_article(word):
    """'an' before a vowel, else 'a' (reference :22-26)."""
    return "an" if word[0].lower() in "aeiou" else "a"


def _qa(question, answer):
    return {"question": question, "answer": answer}


def _lower_label(label):
    # 'Right lung' -> 'right lung'; acronyms ('SVC') and already-lower labels stay (reference :196-197, :277-278)
    return label.lower() if (label[0].isupper() and not label.isupper()) else label


def generate_instruction_report_generation(text, german_suffixe=False):
    """One draw: the question; the answer is the report itself (reference :120-164)."""
    qs = TEMPLATES["report_generation"]["questions"]
    if german_suffixe:
        qs = [q + " in German" for q in qs]
    return _qa(random.choice(qs), f"{text}")


def generate_instruction_phrase_location(bounding_boxes, label):
    """Two draws (question, answer): where is this sentence of the report (reference :167-203)."""
    t = TEMPLATES["phrase_location"]
    boxes = format_boxes(bounding_boxes)
    label = _lower_label(label)
    q = random.choice(t["question_variations"]).format(label)
    return _qa(q, random.choice(t["answer_variations"]).format(boxes))


def generate_instruction_location(bounding_boxes, label):
    """Two draws (question, answer): where is this anatomical region (reference :208-284)."""
    t = TEMPLATES["location"]
    boxes = format_boxes(bounding_boxes)
    label = _lower_label(label)
    q = random.choice(t["questions_variations"]).format(label)
    return _qa(q, random.choice(t["answer_variations"]).format(label, boxes))


def generate_instruction_abnormalities_grouped(bounding_boxes, abnormalities):
    """Two draws (question, then the no-finding sentence or the answer prefix): abnormalities with their boxes, boxes of a repeated
    abnormality grouped under its first mention (reference :288-378)."""
    t = TEMPLATES["abnormalities_grouped"]
    q = random.choice(t["question_variations"])
    if not bounding_boxes or not abnormalities:
        return _qa(q, random.choice(t["no_lesions_answers"]))
    if len(bounding_boxes) != len(abnormalities):
        raise ValueError("Bounding boxes and abnormalities lists must be of equal length.")
    grouped = OrderedDict()
    for name, box in zip(abnormalities, bounding_boxes):
        grouped.setdefault(name, []).append(box)
    parts = [f"{select_article(name)} {name.lower()} located at the coordinates {format_boxes(boxes)}" for name, boxes in grouped.items()]
    return _qa(q, f"{random.choice(t['answer_prefix_variations'])} {'; '.join(parts)}.")


def generate_instruction_foreign_objects(bounding_boxes):
    """Two draws (question, then the no-object sentence or the answer prefix) (reference :381-446)."""
    t = TEMPLATES["foreign_objects"]
    q = random.choice(t["question_variations"])
    if len(bounding_boxes) == 0:
        return _qa(q, random.choice(t["no_objects_answers"]))
    return _qa(q, f"{random.choice(t['answer_prefix_variations'])} {format_boxes(bounding_boxes)}.")


def generate_instruction_abnormalities(abnormalities):
    """Two draws, the ANSWER first and the question last (reference :452-529): presence of abnormalities, each named once in order of
    first mention, lower-cased."""
    t = TEMPLATES["abnormalities"]
    if not abnormalities:
        answer = random.choice(t["no_abnormalities_answers"])
    else:
        names = [n.lower() for n in OrderedDict.fromkeys(abnormalities)]
        if len(names) > 1:
            listing, prefixes = ", ".join(names[:-1]) + " and " + names[-1], t["answer_variations"]
        else:
            listing, prefixes = names[0], t["answer_variations_single"]
        answer = f"{random.choice(prefixes)} {listing}."
    return _qa(random.choice(t["question_variations"]), answer)
