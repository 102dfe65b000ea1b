"""`clip_grad_norm_` and `SGD` for the drop-in path: text.py:385-387 as three streaming launches over the flat buffers.

The reference's loop (text.py:379-387) is `vae.loss(...)`, `loss.mean().backward()`, `torch.nn.utils.clip_grad_norm_(vae.parameters(),
clip_grad)`, `enc_optimizer.step()`.  On the drop-in modules the first two run the HIP path; after `zero_grad()` their backward leaves
every `.grad` as a view of the module's flat gradient buffer (engine.FlatBuffer.deliver_grads).  The two functions here are the other
two lines, same call shape and semantics as torch's, running over those flat buffers:

    from vae_lagging_encoder_amd import optim as lvae_optim
    enc_optimizer = lvae_optim.SGD(vae.encoder.parameters(), lr=1.0)          # text.py:325
    ...
    lvae_optim.clip_grad_norm_(vae.parameters(), clip_grad)                  # text.py:385: norm over encoder AND decoder gradients
    enc_optimizer.step()                                                     # text.py:387

Wherever the gradients are NOT the flat views (another model, gradient accumulation, a parameter the engines do not own) both fall
back to torch's own implementation on the tensors they were given -- same result, more launches.
"""
import torch

from . import engine as _eng
from .engine import P


def _owner_flats(params):
    """The engines' flat buffers that own exactly these parameters, whole modules at a time, with .grad = the flat views; None
    if any parameter is outside that arrangement."""
    params = [p for p in params if p is not None]
    by_id = {}
    for p in params:
        eng = getattr(p, "_lvae_engine", None)
        if eng is None or eng.flat is None or not eng.flat.bound():
            return None
        by_id.setdefault(id(eng), (eng, []))[1].append(p)
    flats = []
    for eng, ps in by_id.values():
        f = eng.flat
        if len(ps) != len(f.params) or {id(q) for q in ps} != {id(q) for q in f.params} or not f.grads_are_views():
            return None
        flats.append(f)
    return flats


_SCAL = {}


def _scalars(device):
    t = _SCAL.get(device)
    if t is None:
        lib = _eng.backend_for(device)
        t = (torch.zeros(4, dtype=torch.float32, device=device),
             torch.empty(lib.lv_sumsq_workspace_floats(), dtype=torch.float32, device=device))
        _SCAL[device] = t
    return t


def clip_grad_norm_(parameters, max_norm, norm_type=2.0):
    """torch.nn.utils.clip_grad_norm_ (text.py:385, image.py:312): total L2 norm over all the gradients, every gradient scaled in
    place by min(1, max_norm / (norm + 1e-6)); returns the norm as a 0-dim device tensor (no host sync).  Over the engines' flat
    gradient buffers this is one reduction per buffer + one scaling pass per buffer, whatever the number of parameters."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    params = [p for p in parameters if p.grad is not None]
    flats = _owner_flats(params) if float(norm_type) == 2.0 and params else None
    if not flats:
        return torch.nn.utils.clip_grad_norm_(params, max_norm, norm_type)
    dev = flats[0].device
    lib, s = _eng.backend_for(dev), _eng.stream_ptr(dev)
    scal, ws = _scalars(dev)            # [0] sumsq, [1] coef, [2] norm
    if len(flats) == 2:
        lib.lv_clip_norm2_f32(P(flats[0].grad), flats[0].numel, P(flats[1].grad), flats[1].numel, P(ws), float(max_norm),
                              P(scal, 0), P(scal, 1), P(scal, 2), s)
    else:
        for i, f in enumerate(flats):
            lib.lv_sumsq_f32(P(f.grad), f.numel, P(ws), P(scal, 0), 1 if i else 0, s)
        lib.lv_clip_coef_f32(P(scal, 0), float(max_norm), P(scal, 1), P(scal, 2), s)
    for f in flats:
        lib.lv_scale_f32(P(f.grad), f.numel, P(scal, 1), s)
    return scal[2].clone()


class SGD(torch.optim.SGD):
    """optim.SGD(params, lr, momentum=0) of text.py:325-326 -- `p -= lr * p.grad` -- as one streaming launch over a module's flat
    buffers when the parameters are exactly one engine's and their gradients the flat views; anything else (momentum, weight decay,
    foreign parameters) is torch's own step.  `param_groups[0]["lr"]` is read at every step, as the reference's lr decay writes it
    (text.py:470-480 rebuilds the optimizer; schedulers that edit the group work too)."""

    def __init__(self, params, lr=1.0, **kw):
        super().__init__(params, lr=lr, **kw)
        self._lr_dev = {}

    @torch.no_grad()
    def step(self, closure=None):
        plain = all(g["momentum"] == 0 and g["weight_decay"] == 0 and g["dampening"] == 0 and not g["nesterov"] and not g.get("maximize", False)
                    for g in self.param_groups)
        if closure is not None or not plain or len(self.param_groups) != 1:
            return super().step(closure)
        group = self.param_groups[0]
        flats = _owner_flats([p for p in group["params"] if p.grad is not None])
        if not flats or sum(len(f.params) for f in flats) != len(group["params"]):
            return super().step()
        for f in flats:
            dev = f.device
            lib, s = _eng.backend_for(dev), _eng.stream_ptr(dev)
            key = (dev, float(group["lr"]))
            t = self._lr_dev.get(key)
            if t is None:
                self._lr_dev = {key: torch.tensor([float(group["lr"]), 1.0], dtype=torch.float32, device=dev)}
                t = self._lr_dev[key]
            lib.lv_sgd_step_f32(P(f.data), P(f.grad), f.numel, P(t, 0), P(t, 1), 0, s)       # coef 1: the clip already scaled the gradient
            f.params[0]._lvae_engine.wgen += 1        # a raw-pointer update is invisible to torch's version counters (engine.weights_version)
        return None
