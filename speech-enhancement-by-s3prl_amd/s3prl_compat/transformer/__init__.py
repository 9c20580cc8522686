"""S3PRL module path shim -> speech_enhancement_by_s3prl_amd (see ../README.md)."""
