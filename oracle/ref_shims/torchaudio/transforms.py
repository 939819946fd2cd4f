class FrequencyMasking:
    def __init__(self, *a, **k):
        raise RuntimeError("shim: FrequencyMasking is not used by the MAT-SED configs (choice[1] == 0)")
