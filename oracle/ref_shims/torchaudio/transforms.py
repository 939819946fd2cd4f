"""Stand-in for torchaudio==2.0.1 `transforms.FrequencyMasking` (authoring container only; torchaudio is not installed).

Written from the package's published definition: `_AxisMasking.forward` applies `functional.mask_along_axis_iid` only when
`iid_masks` is set AND the input is 4-D; otherwise `functional.mask_along_axis(specgram, mask_param, mask_value, axis, p)`, which draws
`value = rand(1) * mask_param`, `min_value = rand(1) * (size(axis) - value)` and fills [long(min_value), long(min_value) + long(value))
along `axis` with `mask_value` for the whole (batch-packed) tensor.  FrequencyMasking uses axis 1.  Parity against the real package is
unpinned (flagged in oracle/matsed_oracle.py and DESIGN.md)."""
import torch


def mask_along_axis(specgram, mask_param, mask_value, axis, p=1.0):
    if axis not in (1, 2):
        raise ValueError("Only Frequency and Time masking are supported")
    if p != 1.0:
        mask_param = min(mask_param, int(specgram.shape[axis] * p))
    if mask_param < 1:
        return specgram
    shape = specgram.size()
    specgram = specgram.reshape([-1] + list(shape[-2:]))
    value = torch.rand(1) * mask_param
    min_value = torch.rand(1) * (specgram.size(axis) - value)
    mask_start = (min_value.long()).squeeze()
    mask_end = (min_value.long() + value.long()).squeeze()
    mask = torch.arange(0, specgram.shape[axis], device=specgram.device, dtype=specgram.dtype)
    mask = (mask >= mask_start) & (mask < mask_end)
    if axis == 1:
        mask = mask.unsqueeze(-1)
    specgram = specgram.masked_fill(mask, mask_value)
    return specgram.reshape(shape[:-2] + specgram.shape[-2:])


class FrequencyMasking(torch.nn.Module):
    def __init__(self, freq_mask_param, iid_masks=False):
        super().__init__()
        self.mask_param, self.axis, self.iid_masks, self.p = freq_mask_param, 1, iid_masks, 1.0

    def forward(self, specgram, mask_value=0.0):
        if self.iid_masks and specgram.dim() == 4:
            raise RuntimeError("shim: the iid branch needs 4-D input, which data_aug.feature_transformation never passes")
        return mask_along_axis(specgram, self.mask_param, mask_value, self.axis, p=self.p)
