from radvlm_amd.data.create_instructions import *  # noqa: F401,F403
from radvlm_amd.data.create_instructions import (create_json_cell_llava, format_boxes, generate_llava_dataset_from_instruction_dataset,  # noqa: F401
                                                 select_article)
# The reference module's star-export surface (radvlm/data/create_instructions.py:1-6): radvlm/data/datasets.py:24 does
# `from radvlm.data.create_instructions import *` and takes `defaultdict` (used by MS_CXR, datasets.py:1088), `Counter`, `np`, `random`,
# `DataLoader` and `custom_collate_fn` from it -- with this module first on the path they must keep coming from here.
import random  # noqa: E402,F401
from collections import Counter, defaultdict  # noqa: E402,F401

import numpy as np  # noqa: E402,F401

try:
    from torch.utils.data import DataLoader  # noqa: E402,F401
except ImportError:  # pragma: no cover
    pass
try:  # the reference's own radvlm.data.utils, found through the extended package path (absent when only this build is installed)
    from radvlm.data.utils import custom_collate_fn  # noqa: E402,F401
except ImportError:
    pass
