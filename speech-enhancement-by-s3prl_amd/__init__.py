"""MI355X-native speech-enhancement hot path behind the S3PRL plugin surface.

Import as ``speech_enhancement_by_s3prl_amd`` (the importable alias package at the repo root extends
its ``__path__`` to this directory, whose name is not a valid Python identifier).
"""
