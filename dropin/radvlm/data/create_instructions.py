from radvlm_amd.data.llava_format import create_json_cell_llava, generate_llava_dataset_from_instruction_dataset  # noqa: F401
