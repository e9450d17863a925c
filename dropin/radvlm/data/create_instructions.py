from radvlm_amd.data.create_instructions import *  # noqa: F401,F403
from radvlm_amd.data.create_instructions import (create_json_cell_llava, format_boxes, generate_llava_dataset_from_instruction_dataset,  # noqa: F401
                                                 select_article)
