"""Deterministic weights for the model-level fixture (tests/golden/make_model_golden.py and tests/test_gpu_model.py):
a ResNet-18 pose net is 11 M parameters -- too large to commit -- so both sides rebuild the SAME weights from the
parameter names: every tensor of the state dict is drawn from its own torch CPU generator seeded by the CRC of its key
(bit-reproducible across machines for a given torch), He-style for convolutions so activations stay O(1)."""
import math
import zlib

import torch


def deterministic_state_dict(state_dict):
    """-> new state dict with the same keys / shapes / dtypes."""
    out = {}
    for key in sorted(state_dict):
        ref = state_dict[key]
        g = torch.Generator().manual_seed(zlib.crc32(key.encode()) & 0x7FFFFFFF)
        leaf = key.rsplit(".", 1)[-1]
        prefix = key[: -len(leaf) - 1]
        if leaf == "num_batches_tracked":
            out[key] = torch.zeros_like(ref)
            continue
        shape = tuple(ref.shape)
        if leaf == "running_mean":
            t = torch.randn(shape, generator=g) * 0.05
        elif leaf == "running_var":
            t = torch.rand(shape, generator=g) * 0.4 + 0.8
        elif leaf == "weight" and len(shape) == 4:
            if "deconv" in key:                      # ConvTranspose2d (in, out, 4, 4), stride 2: in * 4 taps per output
                std = math.sqrt(2.0 / (shape[0] * shape[2] * shape[3] / 4.0))
            else:                                    # Conv2d (out, in, kh, kw)
                std = math.sqrt(2.0 / (shape[1] * shape[2] * shape[3]))
            if "final_layer" in key:
                std *= 4.0                           # (a little contrast in the heat maps)
            t = torch.randn(shape, generator=g) * std
        elif leaf == "weight":                       # batch-norm gamma (incl. the layer's zero-initialised bn: NOT zero here)
            t = torch.rand(shape, generator=g) * 0.4 + 0.8
            if key.startswith("deconv_layers.7."):   # the head's last BN: bring the features the epipolar layer sees to
                t = t * 0.02                         # O(1) (random He weights + unit-variance running stats grow ~100x)
        elif leaf == "bias":
            is_conv = (prefix + ".weight") in state_dict and state_dict[prefix + ".weight"].dim() == 4
            t = torch.randn(shape, generator=g) * (0.01 if is_conv else 0.05)
        else:
            raise KeyError("unexpected state-dict entry %s" % key)
        out[key] = t.to(ref.dtype)
    return out
