"""`from transformer.model import TransformerConfig, TransformerSpecPredictionHead` (model.py:5)."""
from speech_enhancement_by_s3prl_amd.transformer import TransformerConfig, TransformerSpecPredictionHead, TransformerModel  # noqa: F401
