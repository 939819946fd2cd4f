def downsample_avg(*args, **kwargs):
    raise RuntimeError("shim: not on the MAT-SED path")
