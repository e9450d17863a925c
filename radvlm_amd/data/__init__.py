"""The ``radvlm.data`` instruction-dataset contract feeding the training path."""
from .llava_format import create_json_cell_llava, generate_llava_dataset_from_instruction_dataset, mix_datasets  # noqa: F401
