"""Omniglot on-disk format of the reference (image.py:204-205): one `torch.save`d triple (x_train, x_val, x_test) of
float tensors (N, 1, 28, 28) with pixel intensities in [0, 1] -- they are dynamically binarised per batch
(torch.bernoulli, image.py:287,318; AggressiveImageTrainer.binarize here)."""
import torch


def load_image_triple(path, device=None):
    """-> (x_train, x_val, x_test), moved to `device` when given (image.py:204-209)."""
    all_data = torch.load(path, map_location="cpu")
    if not (isinstance(all_data, (tuple, list)) and len(all_data) == 3):
        raise ValueError("%s does not hold an (x_train, x_val, x_test) triple" % path)
    out = []
    for t in all_data:
        t = torch.as_tensor(t).float()
        if t.dim() == 3:
            t = t.unsqueeze(1)
        out.append(t.to(device) if device is not None else t)
    return tuple(out)
