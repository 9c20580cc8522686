"""`from transformer.nn_transformer import TRANSFORMER` (run_downstream.py:19, model.py:4)."""
from speech_enhancement_by_s3prl_amd.transformer import TRANSFORMER  # noqa: F401
