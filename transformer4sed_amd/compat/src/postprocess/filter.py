from transformer4sed_amd.filter import median_filter_torch  # noqa: F401
