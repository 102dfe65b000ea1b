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


def sampler_order(n):
    """The order a shuffled DataLoader pass yields (what image.py:219-221's loaders draw on every `for datum in loader`), with
    the same consumption of torch's global generator: the DataLoader iterator first draws its `_base_seed` (one int64 from the
    global generator, used only for worker seeding -- torch >= 2.0's _BaseDataLoaderIter), THEN RandomSampler draws its own
    int64 seed and takes randperm(n) from a private generator seeded with it.  (tests/test_host_logic.py compares this against a
    real DataLoader under the same seed, orders and the state the global generator is left in.)"""
    torch.empty((), dtype=torch.int64).random_()          # _BaseDataLoaderIter._base_seed: drawn and not used in-process
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    g = torch.Generator()
    g.manual_seed(seed)
    return torch.randperm(n, generator=g).tolist()


class ShuffledLoader(object):
    """`DataLoader(TensorDataset(x, y), batch_size, shuffle=True)` of image.py:215-221 for a tensor that already lives on the
    device: every iteration pass draws a fresh order (order_fn(n) -> list of indices; default `sampler_order`) and yields
    (batch, None) pairs in chunks of batch_size, the last one short (drop_last = False)."""

    def __init__(self, x, batch_size, order_fn=None):
        self.x, self.batch_size = x, int(batch_size)
        self.order_fn = order_fn if order_fn is not None else sampler_order

    def __len__(self):
        return (int(self.x.shape[0]) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        order = list(self.order_fn(int(self.x.shape[0])))
        idx = torch.as_tensor(order, dtype=torch.int64, device=self.x.device)
        for a in range(0, len(order), self.batch_size):
            yield self.x[idx[a:a + self.batch_size]], None
