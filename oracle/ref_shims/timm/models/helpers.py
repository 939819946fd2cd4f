def load_pretrained(*args, **kwargs):
    raise RuntimeError("shim: no pretrained weights available offline")
