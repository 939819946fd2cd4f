from transformer4sed_amd.passt_sed import SEDModel  # noqa: F401
