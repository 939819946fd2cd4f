from transformer4sed_amd.frontend import PasstFeatureExtractor  # noqa: F401
