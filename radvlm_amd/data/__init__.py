"""The ``radvlm.data`` instruction-dataset contract feeding the training path."""
from .create_instructions import (format_boxes, generate_instruction_abnormalities, generate_instruction_abnormalities_grouped,  # noqa: F401
                                  generate_instruction_foreign_objects, generate_instruction_location,
                                  generate_instruction_phrase_location, generate_instruction_report_generation, select_article)
from .llava_format import create_json_cell_llava, generate_llava_dataset_from_instruction_dataset, mix_datasets  # noqa: F401
