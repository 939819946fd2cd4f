from transformer4sed_amd.data_aug import mixup, frame_shift, feature_transformation  # noqa: F401
