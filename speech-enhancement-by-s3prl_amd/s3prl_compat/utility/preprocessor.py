"""`from utility.preprocessor import OnlinePreprocessor` (run_downstream.py:18, runner.py:23, model.py:3, sampler.py:24)."""
from speech_enhancement_by_s3prl_amd.preprocessor import OnlinePreprocessor  # noqa: F401
