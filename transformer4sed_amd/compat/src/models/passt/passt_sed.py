from transformer4sed_amd.passt_sed import PaSST_SED  # noqa: F401  (HIP drop-in for src/models/passt/passt_sed.py)
