from transformer4sed_amd.scheduler import ExponentialDown, update_ema  # noqa: F401
