"""Microbenchmark (measurement tooling): the embedding-gradient scatter on uniformly drawn tokens (the bench's synthetic batches)
and on Zipf-distributed ones (natural text: the most frequent token of a 6368-token batch occurs a few hundred times)."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from vae_lagging_encoder_amd import _lib
from vae_lagging_encoder_amd.engine import P, stream_ptr
dev = torch.device("cuda:0"); s = stream_ptr(dev)
if os.environ.get("LVAE_PROBE_LIB"):
    import ctypes
    lib = _lib.bind(ctypes.CDLL(os.environ["LVAE_PROBE_LIB"]), os.environ["LVAE_PROBE_LIB"])
else:
    lib = _lib.load()
T, B, ni, V = 199, 32, 512, 20001
rs = np.random.RandomState(1)
ranks = np.arange(1, V + 1, dtype=np.float64)
for name, ids in (("uniform", rs.randint(0, V, size=(B, T + 1))),
                  ("zipf(1.0)", rs.choice(V, size=(B, T + 1), p=(1 / ranks) / (1 / ranks).sum()))):
    ids = torch.from_numpy(ids.astype(np.int64)).to(dev)
    rows = torch.empty(T * B, dtype=torch.int32, device=dev); toks = torch.empty_like(rows); tmp = torch.empty(2 * T * B, dtype=torch.int32, device=dev)
    lib.lv_token_sort(P(ids), T + 1, T, B, V, P(rows), P(toks), P(tmp), s)
    dX = torch.randn(T * B, ni, device=dev); dE = torch.empty(V, ni, device=dev)
    cnt = torch.bincount(ids[:, :T].reshape(-1), minlength=V)
    def run(): lib.lv_embed_scatter_full_f32(P(dX), None, 1.0, P(rows), P(toks), T, B, P(dE), ni, V, -1, s)
    for _ in range(3): run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    print("%-10s distinct tokens %5d, most frequent occurs %4d times: %.1f us per scatter" % (name, int((cnt > 0).sum()), int(cnt.max()), e0.elapsed_time(e1) * 1e3 / 20))
