"""`from downstream.model import dummy_upstream` (run_downstream.py:20)."""
from speech_enhancement_by_s3prl_amd.transformer import dummy_upstream  # noqa: F401
