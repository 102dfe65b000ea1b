"""Host-side sequencing of the gfx950 kernels for the Omniglot VAE (ResNet encoder + PixelCNN decoder).

Mirrors what autograd does for the reference's ResNetEncoderV2.forward (modules/encoders/enc_resnet_v2.py:120-126)
and PixelCNNDecoderV2.reconstruct_error (modules/decoders/dec_pixelcnn_v2.py:172-195) with a small explicit tape:
every op below launches hand-written HIP kernels through the C ABI (include/lvae.h) and records the closure that
launches its backward kernels.  Activations are NHWC ([N*H*W][C]); convolutions are im2col + MFMA GEMM with tap
skipping for the masked convolutions; BatchNorm (train) + residual + ELU are one fused pass.
"""
import torch

from . import _lib
from . import engine as _eng
from .engine import P, FlatBuffer, _gemm, stream_ptr


class Act(object):
    """An NHWC activation: t is a contiguous [N*H*W, C] fp32 tensor."""
    __slots__ = ("t", "N", "H", "W", "C", "needs_grad", "bn_nblk", "uses", "bn_src", "fused")

    def __init__(self, t, N, H, W, C, needs_grad=True):
        self.t, self.N, self.H, self.W, self.C, self.needs_grad = t, N, H, W, C, needs_grad
        self.bn_nblk = 0       # > 0: the producer left this many BatchNorm stage-1 partial blocks in the tape's BN workspace
        self.uses = 0          # tape operations that read this activation (a BatchNorm output with ONE reader, a direct or pointwise
        self.bn_src = None     # convolution, has stage 1 of its backward done in that convolution's data-gradient epilogue:
        self.fused = None      # bn_src = (BN input Act, mean, invstd, act) set by Tape.bn; fused = (dv, partials, blocks) by the reader)

    @property
    def P(self):
        return self.N * self.H * self.W


_BN_MAX_BLOCKS = 1024         # partial blocks a BN workspace holds (lv_bn_workspace_floats)
CONV_TERMS = {"f32": 0, "bf16x3": 3, "bf16": 1}      # precision -> terms of lv_conv32_b16 / lv_conv32_wgrad_b16 (0: the exact-f32 entries)


def pack_conv32(lib, s, ent):
    """(Re)build the packed forward / data-gradient weight images of one direct convolution; MaskedConv2d's in-place
    weight.data.mul_(mask) (dec_pixelcnn_v2.py:29, G5) happens here, once per weight version."""
    if ent["mask"] is not None:
        lib.lv_mul_inplace_f32(P(ent["weight"]), P(ent["mask"]), ent["weight"].numel(), s)
    pack = lib.lv_conv32_pack_b16 if ent.get("terms", 0) else lib.lv_conv32_pack_f32      # (hi, lo) bf16 fragments / f32 fragments
    pack(P(ent["weight"]), P(ent["wp"]), ent["k"], ent["nt"], 0, s)
    pack(P(ent["weight"]), P(ent["wpt"]), ent["k"], ent["nt"], 1, s)


class Tape(object):
    def __init__(self, device, precision="f32", train=True, wcache=None, wver=None, fuse_bn_bwd=True):
        self.device = torch.device(device)
        self.lib = _eng.backend_for(self.device)
        # "f32": exact-f32 matrix pipe everywhere.  "bf16x3": the direct 32 -> 32 convolutions with every operand split into two
        # bf16 numbers (three bf16 MFMAs per product, f32 accumulation: f32-like results), the rest exact.  "bf16": plain bf16
        # operands in the direct convolutions and in the im2col GEMMs.
        assert precision in CONV_TERMS
        self.precision = precision
        self.gemm_prec = "bf16" if precision == "bf16" else "f32"
        self.conv_terms = CONV_TERMS[precision]
        # stage 1 of a BatchNorm's backward in the epilogue of the data-gradient convolution in front of it (_fused_bn_bwd_ok);
        # an engine attribute (fuse_bn_bwd), off = the two-launch BatchNorm backward everywhere
        self.fuse_bn_bwd = bool(fuse_bn_bwd)
        self.train = train
        self.back = []
        self.grads = {}
        self._bn_ws = {}
        self.bn_seen = []              # num_batches_tracked buffers of the BatchNorms run in train mode (bumped once, together)
        self.wgrad_pending = []        # (partials, gradient view, outputs, parts, k*k) of the direct convolutions: reduced together
        # packed weight images of the direct convolutions, kept across steps by the owning engine and rebuilt when its
        # weights change (the decoder is frozen during the aggressive inner loop, image.py:300-327)
        self.wcache = wcache if wcache is not None else {}
        self.wver = wver

    # -- helpers -------------------------------------------------------------------------------------------------
    def s(self):
        return stream_ptr(self.device)

    def f32(self, *shape):
        return torch.empty(*shape, dtype=torch.float32, device=self.device)

    def bn_ws(self, C):
        w = self._bn_ws.get(C)
        if w is None:
            w = self.f32(self.lib.lv_bn_workspace_floats(C) + 2 * C)
            self._bn_ws[C] = w
        return w

    def bump_bn_counters(self):
        """BatchNorm2d.num_batches_tracked += 1 for every layer this forward ran in train mode: one fused launch."""
        if self.bn_seen:
            torch._foreach_add_(self.bn_seen, 1)
            self.bn_seen = []

    MAX_PENDING = 4          # summands a consumer can read at once (lv_bn_bwd4_f32)

    def add_grad(self, act, g):
        """Add g (a [P,C] tensor; never written again, so one tensor may serve as the gradient of several activations) to the
        gradient of `act`.  The sum is DEFERRED: up to four summands stay pending, so that a consumer which can read them all
        (BatchNorm backward, lv_bn_bwd4_f32) never needs the 3-pass accumulation launches; a residual add passes its pending
        summands on to both inputs (grad_terms); everybody else gets them summed into a fresh buffer by grad_of()."""
        k = id(act)
        cur = self.grads.get(k)
        if cur is None:
            self.grads[k] = [g]
            return
        if len(cur) == self.MAX_PENDING:
            cur[:] = [self._sum2(cur[0], cur[1])] + cur[2:]
        cur.append(g)

    def _sum2(self, a, b):
        out = torch.empty_like(a)
        self.lib.lv_add_f32(P(a), P(b), P(out), a.numel(), self.s())
        return out

    def grad_of(self, act):
        cur = self.grads.get(id(act))
        if cur is None:
            return None
        while len(cur) > 1:
            cur[:] = [self._sum2(cur[0], cur[1])] + cur[2:]
        return cur[0]

    def grad_terms(self, act):
        """The pending summands of the gradient of `act` (a list of 1..4 tensors in arrival order), or None."""
        return self.grads.get(id(act))

    def backward(self):
        for fn in reversed(self.back):
            fn()
        self.back = []
        self.flush_wgrads()

    def flush_wgrads(self):
        """Sum the weight-gradient partials of every direct convolution of this backward pass in one launch
        (lv_wgrad_reduce_batched): 70 reductions of a few dozen workgroups each otherwise."""
        if not self.wgrad_pending:
            return
        import ctypes
        n = len(self.wgrad_pending)
        desc = (ctypes.c_longlong * (4 * n))()
        for i, (ws, gview, nout, parts, kk) in enumerate(self.wgrad_pending):
            desc[4 * i] = ws.data_ptr()
            desc[4 * i + 1] = gview.data_ptr()
            desc[4 * i + 2] = nout | (parts << 32)
            desc[4 * i + 3] = kk
        self.lib.lv_wgrad_reduce_batched(ctypes.cast(desc, ctypes.c_void_p), n, self.s())
        self.wgrad_pending = []        # (the stream orders the launch before any reuse of the scratch tensors released here)

    # -- ops -----------------------------------------------------------------------------------------------------
    def conv(self, x, weight, gview, stride=1, pad=0, ntaps=None, mask=None, bn_stats=False):
        """nn.Conv2d(bias=False) / MaskedConv2d on NHWC x.  weight: [Cout][Cin][kh][kw] parameter view; gview: where
        its gradient goes.  ntaps: raster-order tap prefix used by forward / data-gradient (None = all taps).
        bn_stats: the output goes straight into a train-mode BatchNorm -- the direct kernels then leave its per-channel
        partial sums in the tape's BN workspace (Act.bn_nblk > 0) and Tape.bn skips its statistics pass."""
        lib, s = self.lib, self.s()
        Cout, Cin, kh, kw = weight.shape
        assert Cin == x.C
        KK = kh * kw
        nt = KK if ntaps is None else ntaps
        Ho = (x.H + 2 * pad - kh) // stride + 1
        Wo = (x.W + 2 * pad - kw) // stride + 1
        Pout = x.N * Ho * Wo
        direct32 = (Cin == 32 and Cout == 32 and stride == 1 and x.H == 28 and x.W == 28 and kh == kw and pad == kh // 2
                    and kh in (3, 5, 7))
        if direct32:
            return self._conv32(x, weight, gview, kh, nt, mask, bn_stats and self.train)
        if KK == 1 and stride == 1 and pad == 0 and mask is None and Cin in (32, 64) and Cout in (32, 64):
            return self._conv1x1(x, weight, gview, bn_stats and self.train)
        x.uses += 1
        if mask is not None:
            # weight.data.mul_(mask) on EVERY forward, eval included (dec_pixelcnn_v2.py:29, G5): the weight gradient spans
            # all taps, so after a decoder update the masked taps are non-zero again until the next forward re-zeroes them
            lib.lv_mul_inplace_f32(P(weight), P(mask), weight.numel(), s)
        y = self.f32(Pout, Cout)
        one_by_one = (KK == 1 and stride == 1 and pad == 0)
        if one_by_one:
            col, wg, ldk = x.t, weight, Cin
        else:
            ldk = KK * Cin
            col = self.f32(Pout, ldk)
            lib.lv_im2col_f32(P(x.t), P(col), ldk, x.N, x.H, x.W, Cin, Ho, Wo, kh, kw, pad, stride, KK, s)
            wg = self.f32(Cout, ldk)
            lib.lv_conv_pack_w_f32(P(weight), P(wg), Cout, Cin, KK, s)
        K = nt * Cin
        _gemm(lib, s, 0, 1, Pout, Cout, K, P(col), ldk, P(wg), ldk, P(y), Cout, prec=self.gemm_prec)
        out = Act(y, x.N, Ho, Wo, Cout)

        def bwd():
            dy = self.grad_of(out)
            if dy is None:
                return
            # weight gradient over ALL taps (masked taps keep non-zero grads in the reference: they enter the clip norm)
            if one_by_one:
                _gemm(lib, s, 1, 0, Cout, Cin, Pout, P(dy), Cout, P(col), ldk, P(gview), Cin, prec=self.gemm_prec)
            else:
                dwg = self.f32(Cout, ldk)
                _gemm(lib, s, 1, 0, Cout, ldk, Pout, P(dy), Cout, P(col), ldk, P(dwg), ldk, prec=self.gemm_prec)
                lib.lv_conv_unpack_dw_f32(P(dwg), P(gview), Cout, Cin, KK, 0, s)
            if x.needs_grad:
                dx = self.f32(x.P, Cin)
                if one_by_one:
                    _gemm(lib, s, 0, 0, Pout, Cin, Cout, P(dy), Cout, P(wg), ldk, P(dx), Cin, prec=self.gemm_prec)
                else:
                    dcol = self.f32(Pout, K)
                    _gemm(lib, s, 0, 0, Pout, K, Cout, P(dy), Cout, P(wg), ldk, P(dcol), K, prec=self.gemm_prec)
                    lib.lv_col2im_f32(P(dcol), K, P(dx), x.N, x.H, x.W, Cin, Ho, Wo, kh, kw, pad, stride, nt, 0, s)
                self.add_grad(x, dx)
        self.back.append(bwd)
        return out

    def _fused_bn_bwd_ok(self, x, nblk):
        """The data gradient wrt x may leave its kernel as the dv of the BatchNorm that produced x: x is a BatchNorm output read by
        this convolution alone, nothing has been added to its gradient, and the partial blocks fit."""
        return (self.fuse_bn_bwd and self.train and x.bn_src is not None and x.uses == 1 and self.grads.get(id(x)) is None
                and 0 < nblk <= _BN_MAX_BLOCKS)

    def _conv32(self, x, weight, gview, k, nt, mask, bn_stats=False):
        """32 -> 32 channel k x k convolution on a 28 x 28 map without an im2col buffer (lv_conv_direct.hip): forward and data
        gradient over the mask's tap prefix, weight gradient over all taps."""
        lib, s = self.lib, self.s()
        x.uses += 1
        ent = self.wcache.get(id(weight))
        terms = self.conv_terms
        if ent is None:
            n = lib.lv_conv32_wpack_floats(nt)
            ent = self.wcache[id(weight)] = dict(ver=object(), wp=self.f32(n), wpt=self.f32(n), weight=weight, k=k, nt=nt, mask=mask,
                                                 terms=terms)
        if self.wver is None or ent["ver"] != self.wver or ent.get("terms", 0) != terms:
            ent["terms"] = terms                                       # (the images of another precision mode are of another format)
            pack_conv32(lib, s, ent)
            ent["ver"] = self.wver if self.wver is not None else object()
        wp, wpt = ent["wp"], ent["wpt"]
        y = self.f32(x.P, 32)
        out = Act(y, x.N, 28, 28, 32)
        with _eng._prof("conv_direct", 2.0 * x.P * 1024 * nt):            # flops over the taps the mask keeps
            stats = bn_stats and lib.lv_conv32_blocks(x.N) <= _BN_MAX_BLOCKS
            if terms:
                lib.lv_conv32_b16(P(x.t), P(wp), P(y), P(self.bn_ws(32)) if stats else None, x.N, k, nt, 0, 0, terms, s)
            elif stats:
                lib.lv_conv32_bnstat_f32(P(x.t), P(wp), P(y), P(self.bn_ws(32)), x.N, k, nt, s)
            else:
                lib.lv_conv32_f32(P(x.t), P(wp), P(y), x.N, k, nt, 0, 0, s)
            if stats:
                out.bn_nblk = lib.lv_conv32_blocks(x.N)

        def bwd():
            dy = self.grad_of(out)
            if dy is None:
                return
            ws = self.f32(lib.lv_conv32_wgrad_ws_floats(x.N, k))
            with _eng._prof("conv_direct", 2.0 * x.P * 1024 * k * k):     # the weight gradient spans all k*k taps (G5)
                if terms:                                                   # stage 1: partials; reduced in flush_wgrads()
                    lib.lv_conv32_wgrad_b16(P(x.t), P(dy), None, P(ws), x.N, k, 0, terms, s)
                else:
                    lib.lv_conv32_wgrad_f32(P(x.t), P(dy), None, P(ws), x.N, k, 0, s)
            self.wgrad_pending.append((ws, gview, k * k * 1024, lib.lv_conv32_wgrad_parts(x.N, k), k * k))
            if x.needs_grad:
                dx = self.f32(x.P, 32)
                nblk = lib.lv_conv32_blocks(x.N)
                with _eng._prof("conv_direct", 2.0 * x.P * 1024 * nt):
                    if self._fused_bn_bwd_ok(x, nblk):
                        # x came out of a BatchNorm (+ ELU) and nobody else reads it: stage 1 of that BatchNorm's backward rides in this
                        # kernel's epilogue (dx leaves as dv, the workgroups leave the partial sums), its reduction launch is gone
                        xin, mean, invstd, act = x.bn_src
                        part = self.f32(nblk * 64)
                        lib.lv_conv32_bnbwd(P(dy), P(wpt), P(dx), P(part), x.N, k, nt, P(x.t), P(xin.t), P(mean), P(invstd), int(act),
                                            terms, s)
                        x.fused = (dx, part, nblk)
                    elif terms:
                        lib.lv_conv32_b16(P(dy), P(wpt), P(dx), None, x.N, k, nt, 1, 0, terms, s)
                    else:
                        lib.lv_conv32_f32(P(dy), P(wpt), P(dx), x.N, k, nt, 1, 0, s)
                self.add_grad(x, dx)
        self.back.append(bwd)
        return out

    def _conv1x1(self, x, weight, gview, bn_stats=False):
        """Pointwise convolution between 32 / 64 channels (lv_conv1x1_*: one pass over the pixels, no split-K)."""
        lib, s = self.lib, self.s()
        x.uses += 1
        Cout, Cin = weight.shape[0], weight.shape[1]
        y = self.f32(x.P, Cout)
        out = Act(y, x.N, x.H, x.W, Cout)
        with _eng._prof("conv_pointwise", 4.0 * x.P * (Cin + Cout)):       # HBM-bound: bytes of the activation in and out
            if bn_stats and lib.lv_conv1x1_blocks(x.P) <= _BN_MAX_BLOCKS:
                lib.lv_conv1x1_bnstat_f32(P(x.t), P(weight), P(y), P(self.bn_ws(Cout)), x.P, Cin, Cout, s)
                out.bn_nblk = int(lib.lv_conv1x1_blocks(x.P))
            else:
                lib.lv_conv1x1_f32(P(x.t), P(weight), P(y), x.P, Cin, Cout, 0, 0, s)

        def bwd():
            dy = self.grad_of(out)
            if dy is None:
                return
            ws = self.f32(lib.lv_conv1x1_wgrad_ws_floats(Cin, Cout))
            with _eng._prof("conv_pointwise", 4.0 * x.P * (Cin + Cout)):
                lib.lv_conv1x1_wgrad_f32(P(x.t), P(dy), None, P(ws), x.P, Cin, Cout, 0, s)  # stage 1: partials; reduced in flush_wgrads()
            self.wgrad_pending.append((ws, gview, Cin * Cout, lib.lv_conv1x1_wgrad_parts(x.P), 0))
            if x.needs_grad:
                # (accumulating into an existing gradient in the kernel's epilogue was measured slower than a separate vectorised add:
                # the read-modify-write of 4-byte pieces costs the pointwise kernel 4 us, the add kernel 3)
                dx = self.f32(x.P, Cin)
                nblk = int(lib.lv_conv1x1_blocks(x.P))
                with _eng._prof("conv_pointwise", 4.0 * x.P * (Cin + Cout)):
                    if self._fused_bn_bwd_ok(x, nblk):         # (see _conv32)
                        xin, mean, invstd, act = x.bn_src
                        part = self.f32(nblk * 2 * Cin)
                        lib.lv_conv1x1_bnbwd_f32(P(dy), P(weight), P(dx), P(part), x.P, Cout, Cin, P(x.t), P(xin.t), P(mean), P(invstd),
                                                 int(act), s)
                        x.fused = (dx, part, nblk)
                    else:
                        lib.lv_conv1x1_f32(P(dy), P(weight), P(dx), x.P, Cout, Cin, 1, 0, s)
                self.add_grad(x, dx)
        self.back.append(bwd)
        return out

    def bn(self, x, bn, g_gamma, g_beta, res=None, act=True):
        """nn.BatchNorm2d (+ residual add) (+ nn.ELU).  Train mode: batch statistics + running-stat update."""
        lib, s = self.lib, self.s()
        x.uses += 1
        if res is not None:
            res.uses += 1
        C, Pn = x.C, x.P
        y = self.f32(Pn, C)
        mean = self.f32(C)
        invstd = self.f32(C)
        nres = 1 if res is not None else 0
        self._bn_prof = _eng._prof("batchnorm", 4.0 * Pn * C * (2 + nres + (0 if (self.train and x.bn_nblk) else 1 if self.train else 0)))
        self._bn_prof.__enter__()
        if self.train and x.bn_nblk:
            # the producing convolution left the per-channel partial sums in the workspace
            lib.lv_bn_fwd_partials_f32(P(x.t), P(bn.weight), P(bn.bias), P(res.t) if res is not None else None, int(act), P(y),
                                       P(mean), P(invstd), P(bn.running_mean), P(bn.running_var), bn.eps, bn.momentum,
                                       P(self.bn_ws(C)), x.bn_nblk, Pn, C, s)
            self.bn_seen.append(bn.num_batches_tracked)
        elif self.train:
            lib.lv_bn_fwd_f32(P(x.t), P(bn.weight), P(bn.bias), P(res.t) if res is not None else None, int(act), P(y),
                              P(mean), P(invstd), P(bn.running_mean), P(bn.running_var), bn.eps, bn.momentum,
                              P(self.bn_ws(C)), Pn, C, s)
            self.bn_seen.append(bn.num_batches_tracked)
        else:
            # eval mode (evaluation passes, sampling: SURVEY.md 8f): the running statistics, one streaming launch
            lib.lv_bn_eval_f32(P(x.t), P(bn.weight), P(bn.bias), P(bn.running_mean), P(bn.running_var), bn.eps,
                               P(res.t) if res is not None else None, int(act), P(y), P(mean), P(invstd), Pn, C, s)
        self._bn_prof.__exit__()
        out = Act(y, x.N, x.H, x.W, C)
        if self.train:
            out.bn_src = (x, mean, invstd, act)

        def bwd():
            terms = self.grad_terms(out)
            if terms is None:
                return
            dx = self.f32(Pn, C)
            if out.fused is not None and len(terms) == 1 and terms[0] is out.fused[0]:
                # stage 1 (dv and the partial sums) came out of the reading convolution's data-gradient kernel: the apply pass alone
                dv, part, nblk = out.fused
                with _eng._prof("batchnorm", 4.0 * Pn * C * 3):
                    lib.lv_bn_bwd_apply_partials_f32(P(x.t), P(dv), P(part), nblk, P(mean), P(invstd), P(bn.weight), P(dx), P(g_gamma),
                                                     P(g_beta), 0, Pn, C, s)
                if res is not None and res.needs_grad:
                    self.add_grad(res, dv)
                if x.needs_grad:
                    self.add_grad(x, dx)
                return
            dys = [P(t_) for t_ in terms] + [None] * (4 - len(terms))
            dv = self.f32(Pn, C)
            # reduce pass: x, y, the summands in, dv out; apply pass: x, dv in, dx out
            with _eng._prof("batchnorm", 4.0 * Pn * C * (6 + len(terms))):
                lib.lv_bn_bwd4_f32(P(x.t), dys[0], dys[1], dys[2], dys[3], P(y), P(mean), P(invstd), P(bn.weight), int(act), P(dv), P(dx),
                                   P(g_gamma), P(g_beta), 0, P(self.bn_ws(C)), Pn, C, s)
            if res is not None and res.needs_grad:
                self.add_grad(res, dv)
            if x.needs_grad:
                self.add_grad(x, dx)
        self.back.append(bwd)
        return out

    def add(self, a, b):
        lib, s = self.lib, self.s()
        a.uses += 1
        b.uses += 1
        y = self.f32(a.P, a.C)
        lib.lv_add_f32(P(a.t), P(b.t), P(y), y.numel(), s)
        out = Act(y, a.N, a.H, a.W, a.C)

        def bwd():
            terms = self.grad_terms(out)
            if terms is None:
                return
            for g in list(terms):                # d(a + b) passes every pending summand on to both inputs, shared, not cloned
                if a.needs_grad:                 # (gradients are never updated in place)
                    self.add_grad(a, g)
                if b.needs_grad:
                    self.add_grad(b, g)
        self.back.append(bwd)
        return out

    def linear(self, x2d, xact, weight, bias, g_w, g_b):
        """y = x W^T + b on a [B, K] matrix.  xact: the Act providing x2d's gradient slot (or None for a leaf)."""
        lib, s = self.lib, self.s()
        if xact is not None:
            xact.uses += 1
        B, K = x2d.shape
        N = weight.shape[0]
        y = self.f32(B, N)
        _gemm(lib, s, 0, 1, B, N, K, P(x2d), K, P(weight), K, P(y), N, add1=P(bias), ld1=0, mod1=1)
        out = Act(y, B, 1, 1, N)

        def bwd():
            dy = self.grad_of(out)
            if dy is None:
                return
            _gemm(lib, s, 1, 0, N, K, B, P(dy), N, P(x2d), K, P(g_w), K)
            lib.lv_colsum_f32(P(dy), N, B, N, P(g_b), None, s)
            if xact is not None and xact.needs_grad:
                dx = self.f32(B, K)
                _gemm(lib, s, 0, 0, B, K, N, P(dy), N, P(weight), K, P(dx), K)
                self.add_grad(xact, dx)
        self.back.append(bwd)
        return out


# ---- network walkers (structure of the reference modules; parameters read from the mirrored nn.Module tree) -----------
def _gv(flat, param):
    """Gradient view in the flat buffer for an nn.Parameter of this module."""
    for n, p in zip(flat.names, flat.params):
        if p is param:
            return flat.gviews[n]
    raise KeyError("parameter not in flat buffer")


def resnet_block(tp, flat, blk, x):
    if blk.downsample is not None:
        r = tp.conv(x, blk.downsample[0].weight, _gv(flat, blk.downsample[0].weight), stride=blk.stride, pad=0)
        residual = tp.bn(r, blk.downsample[1], _gv(flat, blk.downsample[1].weight), _gv(flat, blk.downsample[1].bias), act=False)
    else:
        residual = x
    out = tp.conv(x, blk.conv1.weight, _gv(flat, blk.conv1.weight), stride=blk.stride, pad=1)
    out = tp.bn(out, blk.bn1, _gv(flat, blk.bn1.weight), _gv(flat, blk.bn1.bias), act=True)
    out = tp.conv(out, blk.conv2.weight, _gv(flat, blk.conv2.weight), stride=1, pad=1)
    return tp.bn(out, blk.bn2, _gv(flat, blk.bn2.weight), _gv(flat, blk.bn2.bias), res=residual, act=True)


def encoder_forward(tp, flat, enc, x_img):
    """x_img [B,1,28,28] (any float layout with C=1) -> mulv Act [B, 2nz]."""
    B = x_img.shape[0]
    x = Act(x_img.reshape(B * 28 * 28, 1).contiguous().float(), B, 28, 28, 1, needs_grad=False)
    resnet = enc.main[0]
    for blk in resnet.main:
        x = resnet_block(tp, flat, blk, x)
    conv, bn = enc.main[1], enc.main[2]
    x = tp.conv(x, conv.weight, _gv(flat, conv.weight), stride=1, pad=0)
    x = tp.bn(x, bn, _gv(flat, bn.weight), _gv(flat, bn.bias), act=True)
    return tp.linear(x.t, x, enc.linear.weight, enc.linear.bias, _gv(flat, enc.linear.weight), _gv(flat, enc.linear.bias))


def pixelcnn_block(tp, flat, blk, x):
    m = blk.main
    k = m[3].kernel_size[0]
    h = tp.conv(x, m[0].weight, _gv(flat, m[0].weight), bn_stats=True)
    h = tp.bn(h, m[1], _gv(flat, m[1].weight), _gv(flat, m[1].bias), act=True)
    # type-B mask: taps strictly before the centre in raster order plus the centre itself
    h = tp.conv(h, m[3].weight, _gv(flat, m[3].weight), stride=1, pad=k // 2, ntaps=(k // 2) * k + k // 2 + 1, mask=m[3].mask,
                bn_stats=True)
    h = tp.bn(h, m[4], _gv(flat, m[4].weight), _gv(flat, m[4].bias), act=True)
    h = tp.conv(h, m[6].weight, _gv(flat, m[6].weight), bn_stats=True)
    return tp.bn(h, m[7], _gv(flat, m[7].weight), _gv(flat, m[7].bias), res=x, act=True)


def decoder_forward(tp, flat, dec, x_img, z2d, zact):
    """x_img [B,1,28,28] binarised, z2d [B,nz] -> (logit Act [B*784,1], xflat [B*784])."""
    lib, s = tp.lib, tp.s()
    B = x_img.shape[0]
    npix, fm = 28 * 28, dec.fm_latent
    xflat = x_img.reshape(B * npix).contiguous().float()
    lin = dec.z_transform[0]
    zt = tp.linear(z2d, zact, lin.weight, lin.bias, _gv(flat, lin.weight), _gv(flat, lin.bias))      # [B, fm*784]
    in5_t = tp.f32(B * npix, 1 + fm)
    lib.lv_dec_input_fwd_f32(P(xflat), P(zt.t), P(in5_t), B, npix, fm, s)
    tp.in5 = in5_t                      # (PixelCNNSampler reads the latent maps from here)
    in5 = Act(in5_t, B, 28, 28, 1 + fm)

    def bwd_in():
        d = tp.grad_of(in5)
        if d is None:
            return
        dzt = tp.f32(B, fm * npix)
        lib.lv_dec_input_bwd_f32(P(d), P(dzt), B, npix, fm, s)
        tp.add_grad(zt, dzt)
    tp.back.append(bwd_in)

    pcnn = dec.main[0]
    mA = pcnn.main[0].main
    kA = mA[0].kernel_size[0]
    # MaskABlock: only the image channel is masked (the latent maps see all taps): dense taps, pre-masked weights
    h = tp.conv(in5, mA[0].weight, _gv(flat, mA[0].weight), stride=1, pad=kA // 2, mask=mA[0].mask)
    inp = tp.bn(h, mA[1], _gv(flat, mA[1].weight), _gv(flat, mA[1].bias), act=True)
    direct_inputs = [inp]
    for i in range(1, len(pcnn.main)):
        if i > 2:
            di = direct_inputs.pop(0)
            inp = tp.add(inp, pixelcnn_block(tp, flat, pcnn.direct_connects[i - 3], di))
        inp = pixelcnn_block(tp, flat, pcnn.main[i], inp)
        direct_inputs.append(inp)
    assert len(direct_inputs) == 3
    out = tp.add(inp, pixelcnn_block(tp, flat, pcnn.direct_connects[-1], direct_inputs.pop(0)))
    c1, bn1, c2 = dec.main[1], dec.main[2], dec.main[4]
    h = tp.conv(out, c1.weight, _gv(flat, c1.weight), bn_stats=True)
    h = tp.bn(h, bn1, _gv(flat, bn1.weight), _gv(flat, bn1.bias), act=True)
    logit = tp.conv(h, c2.weight, _gv(flat, c2.weight))
    return logit, xflat


class ImageEncoderEngine(object):
    def __init__(self, module):
        self.m = module
        self.flat = None
        self.precision = "f32"
        self.fuse_bn_bwd = True
        self.gen = 0

    def ensure(self, device):
        device = torch.device(device)
        if self.flat is None or self.flat.device != device or not self.flat.bound():
            self.flat = FlatBuffer(list(self.m.named_parameters()), device)
            self.flat.flat_grads = getattr(self, "_flat_grads_pref", True)      # VAE.use_flat_grads before the buffers existed
        return self.flat

    def forward(self, x_img):
        f = self.ensure(x_img.device)
        self.tape = Tape(x_img.device, self.precision, train=self.m.training, fuse_bn_bwd=self.fuse_bn_bwd)
        self.out = encoder_forward(self.tape, f, self.m, x_img)
        self.tape.bump_bn_counters()
        self.gen += 1
        return self.out.t

    def backward(self, dmulv, gen=None):
        if gen is not None and gen != self.gen:
            raise _lib.LvaeError("encoder activations were overwritten by a later forward()")
        self.tape.add_grad(self.out, dmulv.contiguous().clone())
        self.tape.backward()


class ImageDecoderEngine(object):
    def __init__(self, module):
        self.m = module
        self.flat = None
        self.precision = "f32"
        self.fuse_bn_bwd = True
        self.gen = 0
        self.wgen = 0             # bumped by the fused trainer after a raw-pointer weight update
        self._wcache = {}

    def ensure(self, device):
        device = torch.device(device)
        if self.flat is None or self.flat.device != device or not self.flat.bound():
            self.flat = FlatBuffer(list(self.m.named_parameters()), device)
            self.flat.flat_grads = getattr(self, "_flat_grads_pref", True)      # VAE.use_flat_grads before the buffers existed
        return self.flat

    def weights_version(self):
        f = self.flat
        return (sum(p._version for p in f.params), self.wgen, id(f))

    def refresh_packs(self, device):
        """Bring the cached packed conv weights up to date on the current stream (the fused trainer calls this outside a
        captured region, so that a replayed graph never carries -- or misses -- a repack)."""
        self.ensure(device)
        ver = self.weights_version()
        lib, s = _eng.backend_for(device), stream_ptr(device)
        for ent in self._wcache.values():
            if ent["ver"] != ver:
                pack_conv32(lib, s, ent)
                ent["ver"] = ver

    def forward(self, x_img, z2d):
        """-> rec [B] (BCE summed over pixels)."""
        f = self.ensure(x_img.device)
        wver = self.weights_version()
        tp = Tape(x_img.device, self.precision, train=self.m.training, wcache=self._wcache, wver=wver, fuse_bn_bwd=self.fuse_bn_bwd)
        self.tape = tp
        B = x_img.shape[0]
        self.zact = Act(z2d.contiguous(), B, 1, 1, z2d.shape[1])
        self.logit, self.xflat = decoder_forward(tp, f, self.m, x_img, self.zact.t, self.zact)
        tp.bump_bn_counters()
        self.rec = tp.f32(B)
        tp.lib.lv_sigmoid_bce_fwd_f32(P(self.logit.t), P(self.xflat), P(self.rec), B, 28 * 28, 1e-12, tp.s())
        self.gen += 1
        return self.rec

    def backward(self, drec, gen=None):
        """drec [B] -> parameter grads in self.flat.grad; returns dz [B, nz]."""
        if gen is not None and gen != self.gen:
            raise _lib.LvaeError("decoder activations were overwritten by a later forward()")
        tp = self.tape
        B = self.rec.shape[0]
        dlogit = tp.f32(B * 28 * 28, 1)
        drec_c = drec.contiguous()
        tp.lib.lv_sigmoid_bce_bwd_f32(P(self.logit.t), P(self.xflat), P(drec_c), P(dlogit), B, 28 * 28, 1e-12, tp.s())
        tp.add_grad(self.logit, dlogit)
        tp.backward()
        return tp.grad_of(self.zact)


def _gemm_f32_split(M, N, K, ws_floats):
    """(splits, k tiles per split) lv_gemm_f32 uses for this product (the host logic of lv_gemm_f32.hip restated: the
    pixel-at-a-time sampler sums its fma chain in the same pieces)."""
    BK = 16
    cdiv = lambda a, b: (a + b - 1) // b
    nk = cdiv(K, BK)
    t128 = cdiv(M, 128) * cdiv(N, 128)
    splits = 1
    big = t128 >= 1024
    if not big and ws_floats and t128 >= 48 and nk >= 128:
        s_ = min(cdiv(1024, t128), nk // 32, ws_floats // (M * N))
        if s_ >= 2:
            big, splits = True, s_
    BT = 128 if big else 64
    tiles = cdiv(M, BT) * cdiv(N, BT)
    if not big and ws_floats and tiles < 256 and nk >= 16:
        s_ = min(cdiv(512, tiles), nk // 8, 64, ws_floats // (M * N))
        if s_ > 1:
            splits = s_
    kt = cdiv(max(nk, 1), splits)
    return cdiv(max(nk, 1), kt), kt


class PixelCNNSampler(object):
    """Pixel-at-a-time evaluation of PixelCNNDecoderV2.forward for ancestral sampling (reference
    modules/decoders/dec_pixelcnn_v2.py:201-232; lv_pixelcnn_sample.hip): after `start(z2d)` the image is all zeros and
    `step(i, j)` returns the logits of pixel (i, j) for every image -- the same bits the eval-mode full forward gives at that
    position -- from one launch that extends the cached input maps of the 23 masked convolutions by that position;
    `set_pixel(i, j, values)` then records the drawn pixel.  ~0.1 ms per pixel instead of one 82-convolution forward."""

    def __init__(self, dec_module):
        self.m = dec_module
        self.eng = dec_module._hip

    def _bn_words(self, bn):
        return [bn.weight.data_ptr(), bn.bias.data_ptr(), bn.running_mean.data_ptr(), bn.running_var.data_ptr(), float(bn.eps)]

    def start(self, z2d):
        import ctypes
        import struct
        dec, eng = self.m, self.eng
        dev = z2d.device
        B = z2d.shape[0]
        self.B, self.device = B, dev
        lib = self.lib = _eng.backend_for(dev)
        if dec.training:
            raise _lib.LvaeError("PixelCNNSampler evaluates the decoder in eval mode (running BatchNorm statistics)")
        # one full forward on the empty image: builds the latent maps (in5), brings the packed / masked weight images of the
        # direct convolutions up to date, and re-applies MaskedConv2d's weight.data.mul_(mask) exactly as every forward does
        self.img = torch.zeros(B, 1, 28, 28, dtype=torch.float32, device=dev)
        eng.forward(self.img, z2d.contiguous().float())
        self.in5 = eng.tape.in5
        pcnn = dec.main[0]
        keep = self._keep = []

        def t_(w):            # [Cout][Cin](x1x1) -> ci-major copy
            t = w.detach().reshape(w.shape[0], w.shape[1]).t().contiguous()
            keep.append(t)
            return t.data_ptr()
        nblk = len(pcnn.main) - 1 + len(pcnn.direct_connects)
        self.a1 = torch.zeros(nblk, B, 28 * 28, 32, dtype=torch.float32, device=dev)
        bw = lib.lv_pixelcnn_block_words()

        def block_words(blk, idx):
            m = blk.main
            ent = eng._wcache[id(m[3].weight)]
            k = m[3].kernel_size[0]
            words = [t_(m[0].weight)] + self._bn_words(m[1]) + [ent["wp"].data_ptr(), k, int(ent["nt"])] + self._bn_words(m[4]) + \
                [t_(m[6].weight)] + self._bn_words(m[7]) + [self.a1[idx].data_ptr()]
            assert len(words) == bw
            return words

        def pack(words):
            return b"".join(struct.pack("<d", w) if isinstance(w, float) else struct.pack("<q", int(w)) for w in words)
        mains = [block_words(b_, i) for i, b_ in enumerate(pcnn.main[1:])]
        dcs = [block_words(b_, len(mains) + i) for i, b_ in enumerate(pcnn.direct_connects)]
        raw = b"".join(pack(w) for w in mains + dcs)
        self.tables = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(dev)
        mA = pcnn.main[0].main
        wA = (mA[0].weight.detach() * mA[0].mask)                                      # [64][5][7][7], the mask applied as the forward does
        self.wAt = wA.reshape(64, 5, 49).permute(2, 1, 0).reshape(245, 64).contiguous()
        splits, kt = _gemm_f32_split(B * 784, 64, 245, _eng._gemm_ws(lib, stream_ptr(dev)).numel())
        c1, bnH, c2 = dec.main[1], dec.main[2], dec.main[4]
        self.logit = torch.zeros(B, 28 * 28, dtype=torch.float32, device=dev)
        self.c2 = c2.weight.detach().reshape(64).contiguous()
        words = [self.in5.data_ptr(), self.wAt.data_ptr()] + self._bn_words(mA[1]) + [kt * 16 if splits > 1 else 0,
                 self.tables.data_ptr(), self.tables.data_ptr() + 8 * bw * len(mains), t_(c1.weight)] + self._bn_words(bnH) + \
            [self.c2.data_ptr(), self.logit.data_ptr(), int(lib.lv_conv32_tap_split(B)), len(mains), len(dcs)]
        assert len(words) == lib.lv_pixelcnn_net_words()
        assert splits <= 2
        self.net = (ctypes.c_char * (8 * len(words))).from_buffer_copy(pack(words))
        return self

    def step(self, i, j):
        """-> logits [B] of pixel (i, j) (a view into the sampler's logit map)."""
        import ctypes
        self.lib.lv_pixelcnn_pixel_step_f32(ctypes.cast(self.net, ctypes.c_void_p), self.B, i, j, stream_ptr(self.device))
        return self.logit[:, i * 28 + j]

    def set_pixel(self, i, j, values):
        """values [B] (or [B, 1]) in {0, 1}: the drawn pixel."""
        v = values.reshape(self.B).float()
        self.img[:, 0, i, j] = v
        self.in5.view(self.B, 28 * 28, -1)[:, i * 28 + j, 0] = v
