"""`from downstream.solver import get_optimizer` (runner.py:22, sampler.py:23)."""
from speech_enhancement_by_s3prl_amd.solver import get_optimizer  # noqa: F401
