"""Conv-TasNet masker on the HIP path (mirror of puresound/nnet/conv_tasnet.py:11-377).

The modules keep the reference's parameter tree (so its checkpoints load with identical keys) but
`forward` packs the weights once into the kernel-side layout ("plan") and enqueues the fused HIP
kernels through the C ABI: per normal TCN block one exact-fp32 MFMA GEMM for in_conv, one depthwise
streaming kernel, one GEMM for the pointwise conv and one GEMM for out_conv + residual, each applying
the previous stage's gLN/gGN/bN1d + PReLU while it loads its input and emitting the partial
statistics the next global norm needs.
"""
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import hip
from ..ops import op_module, same_shape
from .._abi import PS_NORM_AFFINE, PS_NORM_GLOBAL, TcnBlock, ptr
from .lobe.cnn import DepthwiseSeparableConv1d
from .lobe.norm import ChanLN, GlobLN, get_norm, norm_plan


_PLAN_SERIAL = [0]


GEMM_PLANES = {"fp32": 0, "bf16": 1, "fp16x2": 2, "bf16x3": 3}


class _PlanCache:
    """Packed-weight caches hold raw device pointers (ctypes) -- never pickle / deepcopy them."""

    _CACHE_ATTRS = ("_plan", "_plan_sig", "_blocks", "_blocks_sig", "_workspace")

    def __getstate__(self):
        state = dict(self.__dict__)
        for k in self._CACHE_ATTRS:
            if k in state:
                state[k] = None
        return state


def _param_signature(module: nn.Module):
    """Cheap fingerprint of every parameter/buffer: a changed value (in-place edit, load_state_dict,
    .to(device)) changes the version counter or the data pointer, which invalidates the packed plan."""
    sig = []
    for t in list(module.parameters()) + list(module.buffers()):
        sig.append((t.data_ptr(), t._version))
    sig.append(module.training)
    return tuple(sig)


@op_module("tcn_block_fwd", same_shape, cpu="tcn_block")
class TCN(_PlanCache, nn.Module):
    """Input 1x1 conv -> norm -> PReLU -> depthwise-separable conv -> output 1x1 conv -> + residual
    (conv_tasnet.py:11-90)."""

    def __init__(self, in_channels: int, hid_channels: int, kernel: int, dilation: int, dropout: float = 0.0,
                 emb_dim: int = 0, causal: bool = False, tcn_norm: str = "gLN", dconv_norm: str = "gGN") -> None:
        super().__init__()
        self.in_channels, self.hid_channels = in_channels, hid_channels
        self.kernel, self.dilation, self.emb_dim, self.causal = kernel, dilation, emb_dim, causal
        norm = get_norm(tcn_norm)
        self.in_conv = nn.Sequential(
            nn.Conv1d(in_channels + emb_dim, hid_channels, kernel_size=1, bias=False, groups=1),
            norm(hid_channels), nn.PReLU())
        self.dconv = nn.Sequential(
            DepthwiseSeparableConv1d(in_channels=hid_channels, out_channels=hid_channels, hid_channels=None,
                                     kernel=kernel, dilation=dilation, skip=False, causal=causal,
                                     norm_cls=dconv_norm),
            nn.Dropout(p=dropout))
        self.out_conv = nn.Conv1d(hid_channels, in_channels, kernel_size=1, stride=1)
        self._plan = None
        self._plan_sig = None

    # -- kernel-side weight layout ----------------------------------------------------------------
    #: arithmetic of the three 1x1 convs (fp32 tensors, fp32 accumulation in every case):
    #:   "fp16x2" (default)  every fp32 operand as two fp16 terms, three products on the fp16 matrix pipe, operands scaled
    #:                       into fp16's range by powers of two (bounds behind the global norms, the producer's maxima for
    #:                       the residual stream).  Measured against fp64 products and against the reference's golden
    #:                       vectors its error is the exact-fp32 MFMA path's; half the matrix-pipe work of
    #:                       "bf16x3".  Blocks without global norms (bN1d, cLN) run "bf16x3" instead.
    #:   "bf16x3"            three bf16 terms, six products: fp32-accurate by construction (dropped terms <= 2^-24)
    #:   "fp32"              v_mfma_f32 on fp32 operands; a row's result is bit-identical whatever batch it is part of
    #:   "bf16"              operands rounded to bf16: what BASELINE.json names for its bf16 configurations; not an fp32 result
    gemm_precision = "fp16x2"
    #: with gemm_precision "bf16": keep the block's hidden maps (y1, y2, y3 inside the fused driver's workspace) as
    #: bf16 rows -- BASELINE's "bf16" configurations name bf16 storage with fp32 accumulation.  False: fp32 rows.
    hidden_bf16 = True
    #: with gemm_precision "bf16" and hidden_bf16: the residual stream between the blocks of a fused stack is bf16 rows as
    #: well (ps_conv_tasnet_bf16_rows) -- "bf16 storage / fp32 accumulate" for every activation row of the stack
    stream_bf16 = True

    def gemm_planes_for_plan(self) -> int:
        """ps_tcn_block.gemm_planes this block runs with: its gemm_precision, unless the block cannot take it."""
        planes = GEMM_PLANES[self.gemm_precision]
        if max(self.in_channels, self.hid_channels) > 512:
            return 0  # ps_conv1x1_bf16_f32 keeps prologue tables for up to 512 input channels: wider blocks run fp32
        if planes == 2:
            # fp16x2 scales the activations of the pointwise / output convs into fp16's range: behind a global norm (gLN,
            # gGN) by a bound on the normalised values, behind an eval BatchNorm (bN1d: a per-channel affine map) by the
            # maxima the producing kernel measured, mapped through the largest scale and shift.  cLN blocks (run stage by
            # stage) take the three-plane bf16 split.
            dsc = self.dconv[0]
            if not all(isinstance(m, (GlobLN, nn.GroupNorm, nn.BatchNorm1d)) for m in (dsc.depthwise[1], dsc.pointwise[1])):
                return 3
            if isinstance(self.in_conv[1], ChanLN):
                return 3
        return planes

    def plan(self, device: torch.device) -> dict:
        planes = self.gemm_planes_for_plan()
        hb = bool(self.hidden_bf16) and planes == 1 and self.kernel == 3 and 2 * self.dilation + 8 <= 288
        sig = (_param_signature(self), str(device), planes, hb, bool(self.stream_bf16))
        if self._plan is not None and self._plan_sig == sig:
            return self._plan
        if self.training and self.dconv[1].p > 0:
            raise RuntimeError("TCN: dropout is active; the HIP path is inference only -- call .eval()")
        if not self.causal and self.kernel % 2 == 0:
            raise RuntimeError("TCN: an even kernel with symmetric padding changes the length (the reference "
                               "fails at the residual add)")
        c, h = self.in_channels, self.hid_channels
        dsc = self.dconv[0]
        f32 = dict(dtype=torch.float32, device=device)
        w_in = self.in_conv[0].weight.detach().to(**f32)
        t = {}
        t["in_wt"] = hip.pack_wt(w_in[:, :c, 0])
        t["in_embed_w"] = w_in[:, c:, 0].contiguous() if self.emb_dim > 0 else None
        kinds = {}
        for name, mod in (("in", self.in_conv[1]), ("dw", dsc.depthwise[1]), ("pw", dsc.pointwise[1])):
            if isinstance(mod, ChanLN):
                # per-frame LayerNorm over channels: no fused prologue form, the block runs stage by stage
                kinds[name], g, b = "cln", mod.gamma.detach().float(), mod.beta.detach().float()
            else:
                kinds[name], g, b = norm_plan(mod)
            t[name + "_gamma"], t[name + "_beta"] = g.to(**f32).contiguous(), b.to(**f32).contiguous()
        t["in_slope"] = self.in_conv[2].weight.detach().to(**f32).contiguous()
        t["dw_slope"] = dsc.depthwise[2].weight.detach().to(**f32).contiguous()
        t["pw_slope"] = dsc.pointwise[2].weight.detach().to(**f32).contiguous()
        for k, mod in (("in_slope", self.in_conv[2]), ("dw_slope", dsc.depthwise[2]), ("pw_slope", dsc.pointwise[2])):
            if mod.weight.numel() != 1:
                raise NotImplementedError("PReLU with per-channel slopes is not on the HIP path")
        t["dw_w"] = dsc.depthwise[0].weight.detach().to(**f32).contiguous()
        t["dw_b"] = dsc.depthwise[0].bias.detach().to(**f32).contiguous()
        t["pw_wt"] = hip.pack_wt(dsc.pointwise[0].weight.detach().to(**f32))
        t["pw_b"] = dsc.pointwise[0].bias.detach().to(**f32).contiguous()
        t["out_wt"] = hip.pack_wt(self.out_conv.weight.detach().to(**f32))
        t["out_b"] = self.out_conv.bias.detach().to(**f32).contiguous()
        b = TcnBlock()
        b.C, b.H, b.P, b.dilation, b.causal = c, h, self.kernel, self.dilation, int(self.causal)
        fused = "cln" not in kinds.values()
        b.in_norm, b.dw_norm, b.pw_norm = [kinds[k] if fused else 0 for k in ("in", "dw", "pw")]
        b.E = self.emb_dim
        b.gemm_planes = planes
        b.hidden_bf16 = int(hb)
        rows_bf16 = bool(hb and self.stream_bf16)
        if planes == 2:
            assert fused and kinds["dw"] in (PS_NORM_GLOBAL, PS_NORM_AFFINE) and kinds["pw"] in (PS_NORM_GLOBAL, PS_NORM_AFFINE)
            for i, (key, wsrc) in enumerate((("in_wb", w_in[:, :c, 0]),
                                             ("pw_wb", dsc.pointwise[0].weight.detach().to(**f32)),
                                             ("out_wb", self.out_conv.weight.detach().to(**f32)))):
                t[key], b.w_exp[i] = hip.pack_wt_f16x2(wsrc)
            # bound on |PReLU(gamma z + beta)| for the fp16 range: the PReLU runs before the split, and a slope beyond 1
            # in magnitude makes a negative value LARGER -- the factor max(1, |slope|) covers it
            fd = max(1.0, abs(float(t["dw_slope"][0])))
            fp = max(1.0, abs(float(t["pw_slope"][0])))
            b.dw_gmax, b.dw_bmax = float(t["dw_gamma"].abs().max()) * fd, float(t["dw_beta"].abs().max()) * fd
            b.pw_gmax, b.pw_bmax = float(t["pw_gamma"].abs().max()) * fp, float(t["pw_beta"].abs().max()) * fp
        elif planes:
            t["in_wb"] = hip.pack_wt_bf16(w_in[:, :c, 0], planes)
            t["pw_wb"] = hip.pack_wt_bf16(dsc.pointwise[0].weight.detach().to(**f32), planes)
            t["out_wb"] = hip.pack_wt_bf16(self.out_conv.weight.detach().to(**f32), planes)
            if rows_bf16 and fused and kinds["dw"] == PS_NORM_GLOBAL and kinds["pw"] == PS_NORM_GLOBAL:
                # bf16 residual stream: large launches run ps_conv1x1_f16_rows (bf16 rows, one fp16 product) -- the fp16
                # images of the weights and the bounds of the normalised activations, as for "fp16x2"
                for i, (key, wsrc) in enumerate((("in_wf", w_in[:, :c, 0]),
                                                 ("pw_wf", dsc.pointwise[0].weight.detach().to(**f32)),
                                                 ("out_wf", self.out_conv.weight.detach().to(**f32)))):
                    t[key], b.w_exp[i] = hip.pack_wt_f16x2(wsrc)
                fd = max(1.0, abs(float(t["dw_slope"][0])))
                fp = max(1.0, abs(float(t["pw_slope"][0])))
                b.dw_gmax, b.dw_bmax = float(t["dw_gamma"].abs().max()) * fd, float(t["dw_beta"].abs().max()) * fd
                b.pw_gmax, b.pw_bmax = float(t["pw_gamma"].abs().max()) * fp, float(t["pw_beta"].abs().max()) * fp
        for k, v in t.items():
            setattr(b, k, ptr(v))
        _PLAN_SERIAL[0] += 1
        # tensors kept alive alongside the raw pointers
        self._plan = {"block": b, "tensors": t, "serial": _PLAN_SERIAL[0], "kinds": kinds, "fused": fused,
                      "rows_bf16": rows_bf16}
        self._plan_sig = sig
        return self._plan

    def forward_padded_staged(self, x: torch.Tensor, t: int, embed: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Stage-by-stage block for norm mixes the fused driver has no form for (cLN): every 1x1 conv is its own
        ps_conv1x1_f32, cLN + PReLU is ps_chan_layernorm_f32, gLN / bN1d stay consumer prologues.
        `embed` is used as given (already normalised by ConvTasNet)."""
        p = self.plan(x.device)
        w, kinds = p["tensors"], p["kinds"]
        n, c, ldt = x.shape
        h = self.hid_channels
        new = lambda rows: torch.empty(n, rows, ldt, dtype=torch.float32, device=x.device)  # noqa: E731

        def settle(y, stats, name, rows):
            """-> (tensor, prologue) the next stage consumes for the norm + PReLU that follow `y`."""
            if kinds[name] == "cln":
                return hip.chan_layernorm(y, t, w[name + "_gamma"], w[name + "_beta"], 1e-8,
                                          slope=w[name + "_slope"]), None
            return y, hip.make_prologue(kinds[name], True, stats, rows * t, 1e-8, w[name + "_gamma"], w[name + "_beta"],
                                        w[name + "_slope"])

        bias_n = None if embed is None else hip.embed_bias(embed.float(), w["in_embed_w"], False)
        y, st = hip.conv1x1(x, t, w["in_wt"], h, None, None, bias_n, want_stats=kinds["in"] == PS_NORM_GLOBAL,
                            out=new(h))
        a, pro = settle(y, st, "in", h)
        left = (self.kernel - 1) * self.dilation if self.causal else ((self.kernel - 1) // 2) * self.dilation
        y, st = hip.dwconv(a, t, w["dw_w"], w["dw_b"], self.dilation, left, pro, kinds["dw"] == PS_NORM_GLOBAL)
        a, pro = settle(y, st, "dw", h)
        y, st = hip.conv1x1(a, t, w["pw_wt"], h, pro, w["pw_b"], want_stats=kinds["pw"] == PS_NORM_GLOBAL, out=new(h))
        a, pro = settle(y, st, "pw", h)
        out, _ = hip.conv1x1(a, t, w["out_wt"], c, pro, w["out_b"], res=x, out=new(c))
        return out

    def forward(self, x: torch.Tensor, embed: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [N,C,T], embed [N,E] -> [N,C,T] (conv_tasnet.py:67-90)."""
        hip.require_device(x, "TCN.forward")
        if (embed is not None) != (self.emb_dim > 0):
            raise RuntimeError(f"TCN.forward: block built with emb_dim={self.emb_dim} but embed is "
                               f"{'given' if embed is not None else 'missing'} (the reference fails in in_conv)")
        p = self.plan(x.device)
        t = x.shape[-1]
        if not p["fused"]:
            return hip.unpad_rows(self.forward_padded_staged(hip.pad_rows(x), t, embed), t)
        blocks = (TcnBlock * 1)(p["block"])
        out = hip.conv_tasnet(blocks, 1, hip.pad_rows(x), t, self.in_channels, self.hid_channels,
                              None if embed is None else embed.contiguous(), False)
        return hip.unpad_rows(out, t)


@op_module("gated_tcn_fwd", same_shape, cpu="gated_tcn_block")
class GatedTCN(_PlanCache, nn.Module):
    """Gated TCN block (conv_tasnet.py:93-215): 1x1 in_conv; left = PReLU(norm(dense dilated conv)); right =
    sigmoid(PReLU(norm(dense dilated conv of [h; e] or of FiLM(h)))); out_conv(left * right); + residual.

    The dense P-tap convolutions are GEMMs with K = P*H: ps_unfold_taps_f32 lays the P shifted copies of h (with the
    conv's zero padding, the FiLM affine and the repeated embedding rows) side by side and ps_conv1x1_f32 does the
    rest on MFMA, its epilogue producing the gLN statistics the gating kernel needs."""

    def __init__(self, in_channels: int, hid_channels: int, kernel: int, dilation: int, dropout: float = 0.0,
                 emb_dim: int = 0, causal: bool = False, tcn_norm: str = "gLN", use_film: bool = False):
        super().__init__()
        self.in_channels, self.hid_channels, self.kernel, self.dilation, self.emb_dim = \
            in_channels, hid_channels, kernel, dilation, emb_dim
        self.causal = causal
        self.padd = (kernel - 1) * dilation // 2 if not causal else (kernel - 1) * dilation
        self.tcn_norm = tcn_norm
        norm_cls = get_norm(tcn_norm)
        self.use_film = use_film
        self.in_conv = nn.Conv1d(in_channels, hid_channels, kernel_size=1, bias=False, groups=1)
        self.left_conv = nn.Sequential(
            nn.Conv1d(hid_channels, hid_channels, kernel_size=kernel, dilation=dilation, bias=False,
                      padding=self.padd, groups=1),
            norm_cls(hid_channels), nn.PReLU(), nn.Dropout(p=dropout))
        if not self.use_film:
            right_in_dim = hid_channels + emb_dim
        else:
            self.cond_scale = nn.Conv1d(emb_dim, hid_channels, kernel_size=1, bias=False)
            self.cond_bias = nn.Conv1d(emb_dim, hid_channels, kernel_size=1, bias=False)
            right_in_dim = hid_channels
        self.right_conv = nn.Sequential(
            nn.Conv1d(right_in_dim, hid_channels, kernel_size=kernel, dilation=dilation, bias=False,
                      padding=self.padd, groups=1),
            norm_cls(hid_channels), nn.PReLU(), nn.Dropout(p=dropout), nn.Sigmoid())
        self.out_conv = nn.Conv1d(hid_channels, in_channels, kernel_size=1, bias=False, groups=1)
        self._plan = None
        self._plan_sig = None

    @staticmethod
    def _unfolded(w: torch.Tensor) -> torch.Tensor:
        """[M, Kc, P] -> [M, P*Kc] with column j*Kc + k (the row order of ps_unfold_taps_f32)."""
        return w.permute(0, 2, 1).reshape(w.shape[0], -1).contiguous()

    def plan(self, device: torch.device) -> dict:
        sig = (_param_signature(self), str(device))
        if self._plan is not None and self._plan_sig == sig:
            return self._plan
        if self.training and (self.left_conv[3].p > 0 or self.right_conv[3].p > 0):
            raise RuntimeError("GatedTCN: dropout is active; the HIP path is inference only -- call .eval()")
        span = (self.kernel - 1) * self.dilation
        if not self.causal and span % 2:
            raise RuntimeError("GatedTCN: (kernel-1)*dilation is odd, the symmetric padding changes the length "
                               "(the reference fails at the residual add)")
        f32 = dict(dtype=torch.float32, device=device)
        t = dict(w_in=hip.pack_wt(self.in_conv.weight.detach().to(**f32)),
                 w_left=hip.pack_wt(self._unfolded(self.left_conv[0].weight.detach().to(**f32))),
                 w_right=hip.pack_wt(self._unfolded(self.right_conv[0].weight.detach().to(**f32))),
                 w_out=hip.pack_wt(self.out_conv.weight.detach().to(**f32)))
        for side, seq in (("left", self.left_conv), ("right", self.right_conv)):
            if seq[2].weight.numel() != 1:
                raise NotImplementedError("PReLU with per-channel slopes is not on the HIP path")
            t[side + "_slope"] = seq[2].weight.detach().to(**f32).contiguous()
            mod = seq[1]
            if isinstance(mod, ChanLN):
                t[side + "_kind"] = "cln"
                g, b = mod.gamma.detach(), mod.beta.detach()
            else:
                kind, g, b = norm_plan(mod)
                t[side + "_kind"] = kind
            t[side + "_gamma"], t[side + "_beta"] = g.to(**f32).contiguous(), b.to(**f32).contiguous()
        if self.use_film:
            t["w_film"] = torch.cat([self.cond_scale.weight.detach()[:, :, 0], self.cond_bias.weight.detach()[:, :, 0]],
                                    0).to(**f32).contiguous()
        self._plan, self._plan_sig = t, sig
        return t

    def forward_padded(self, x: torch.Tensor, t: int, embed: Optional[torch.Tensor] = None) -> torch.Tensor:
        """padded [N,C,ldt] -> padded [N,C,ldt] (embed is used as given: ConvTasNet normalises it beforehand)."""
        p = self.plan(x.device)
        n, c, ldt = x.shape
        h, k, d = self.hid_channels, self.kernel, self.dilation
        left = self.padd
        # causal with a global norm (the constructor's default norm): the reference pads BOTH sides of its dense convs
        # and trims only after out_conv (conv_tasnet.py:203-211), so gLN's statistics run over T + padding frames, the
        # last `padding` of them convolutions over the zero-padded tail.  Reproduced: the conv / norm / product stages
        # cover tt = T + padding frames, out_conv the first T.
        tt, ld_in = t, ldt
        if self.causal and p["left_kind"] == PS_NORM_GLOBAL:
            tt = t + self.padd
            if hip.padded_frames(tt) > ldt:  # rows with room for the tail
                x = hip.pad_rows(hip.unpad_rows(x, t), tt)
                ldt = x.shape[-1]
        new = lambda rows: torch.empty(n, rows, ldt, dtype=torch.float32, device=x.device)  # noqa: E731
        y, _ = hip.conv1x1(x, t, p["w_in"], h, out=new(h))
        scale = shift = emb_rows = None
        if embed is not None:
            embed = embed.float().contiguous()
            if self.use_film:
                sb = hip.embed_bias(embed, p["w_film"], False)              # [N, 2H]: scale | bias
                scale, shift = sb[:, :h].contiguous(), sb[:, h:].contiguous()
            else:
                emb_rows = embed
        col_l = hip.unfold_taps(y, t, k, d, left, t_out=tt)
        col_r = col_l if (scale is None and emb_rows is None) else hip.unfold_taps(y, t, k, d, left, scale, shift,
                                                                                  emb_rows, t_out=tt)
        cln = p["left_kind"] == "cln"
        want = (not cln) and p["left_kind"] == PS_NORM_GLOBAL
        lo, ls = hip.conv1x1(col_l, tt, p["w_left"], h, want_stats=want, out=new(h))
        ro, rs = hip.conv1x1(col_r, tt, p["w_right"], h, want_stats=want, out=new(h))
        if cln:
            lf = hip.chan_layernorm(lo, t, p["left_gamma"], p["left_beta"], 1e-8, slope=p["left_slope"])
            g = hip.chan_layernorm(ro, t, p["right_gamma"], p["right_beta"], 1e-8, slope=p["right_slope"],
                                   sigmoid=True, mul=lf)
        else:
            pl = hip.make_prologue(p["left_kind"], True, ls, h * tt, 1e-8, p["left_gamma"], p["left_beta"],
                                   p["left_slope"])
            pr = hip.make_prologue(p["right_kind"], True, rs, h * tt, 1e-8, p["right_gamma"], p["right_beta"],
                                   p["right_slope"])
            g = hip.gated_product(lo, ro, tt, pl, pr)
        out, _ = hip.conv1x1(g, t, p["w_out"], c, res=x, out=new(c))
        if ldt != ld_in:  # back to the caller's row length
            out = hip.pad_rows(hip.unpad_rows(out, t), ld_in)
            if out.shape[-1] != ld_in:
                raise RuntimeError("GatedTCN: the caller's rows are shorter than padded_frames(T)")
        return out

    def forward(self, x: torch.Tensor, embed: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [N,C,T], embed [N,E] -> [N,C,T] (conv_tasnet.py:178-215)."""
        hip.require_device(x, "GatedTCN.forward")
        if embed is not None and not self.use_film and self.emb_dim == 0:
            raise RuntimeError("GatedTCN.forward: block built with emb_dim=0 but embed is given "
                               "(the reference fails in right_conv)")
        t = x.shape[-1]
        return hip.unpad_rows(self.forward_padded(hip.pad_rows(x), t, embed), t)


@op_module("conv_tasnet_fwd", same_shape, cpu="conv_tasnet")
class ConvTasNet(_PlanCache, nn.Module):
    """R repeats of X dilated TCN blocks with optional speaker-embedding injection
    (conv_tasnet.py:218-377).  Encoder/decoder live outside, as in the reference."""

    def __init__(self, input_dim: int = 512, embed_dim: int = 256, embed_norm: bool = False,
                 tcn_layer: str = "normal", tcn_kernel: int = 3, tcn_dim: int = 256, tcn_dilated_basic: int = 2,
                 per_tcn_stack: int = 5, repeat_tcn: int = 4, tcn_with_embed: List = [1, 0, 0, 0, 0],
                 tcn_norm: str = "gLN", dconv_norm: str = "gGN", causal: bool = False):
        super().__init__()
        self.input_dim = input_dim
        self.embed_dim = embed_dim
        self.embed_norm = embed_norm
        self.tcn_layer = tcn_layer
        self.tcn_dim = tcn_dim
        self.tcn_kernel = tcn_kernel
        self.per_tcn_stack = per_tcn_stack
        self.repeat_tcn = repeat_tcn
        self.tcn_dilated_basic = tcn_dilated_basic
        self.tcn_with_embed = tcn_with_embed
        self.tcn_norm = tcn_norm
        self.dconv_norm = dconv_norm
        self.causal = causal

        if self.tcn_layer.lower() == "normal":
            tcn_cls = TCN
        elif self.tcn_layer.lower() == "gated":
            tcn_cls = GatedTCN
        else:
            raise NameError
        assert per_tcn_stack == len(tcn_with_embed)
        self.tcn_list = nn.ModuleList()
        for _ in range(repeat_tcn):
            stack = []
            for i in range(per_tcn_stack):
                kw = dict(kernel=tcn_kernel, dilation=tcn_dilated_basic ** i,
                          emb_dim=embed_dim if tcn_with_embed[i] else 0, causal=causal, tcn_norm=tcn_norm)
                if tcn_cls is TCN:
                    kw["dconv_norm"] = dconv_norm
                stack.append(tcn_cls(input_dim, tcn_dim, **kw))
            self.tcn_list.append(nn.ModuleList(stack))
        self._blocks = None
        self._blocks_sig = None
        self._workspace = None

    def set_gemm_precision(self, name: str) -> "ConvTasNet":
        """"fp16x2" (default: two fp16 terms per operand, three products) | "bf16x3" | "fp32" | "bf16" for the 1x1 convs of
        every normal TCN block (see TCN.gemm_precision).  The switch is per module; the process-wide
        set_recurrent_gemm_precision / PS_RECURRENT_GEMM of round 1 no longer exist."""
        if name not in GEMM_PLANES:
            raise ValueError(f"gemm precision must be one of {sorted(GEMM_PLANES)}")
        for stack in self.tcn_list:
            for m in stack:
                m.gemm_precision = name
        return self

    # -- plan: one ps_tcn_block per TCN, in execution order -------------------------------------------
    def block_array(self, device: torch.device):
        if self.tcn_layer.lower() != "normal":
            raise RuntimeError("block_array: only the normal TCN stack runs through ps_conv_tasnet_f32")
        mods = [m for stack in self.tcn_list for m in stack]
        plans = [m.plan(device) for m in mods]  # each TCN re-validates its own fingerprint
        sig = tuple(p["serial"] for p in plans)
        if self._blocks is None or self._blocks_sig != sig:
            self._blocks = (TcnBlock * len(plans))(*[p["block"] for p in plans])
            self._blocks_sig = sig
        return self._blocks, len(plans)

    #: forward_padded takes `x_amax` (per-utterance bounds on |x_pad|) for blocks in the fp16x2 arithmetic
    takes_input_range = True

    def forward_padded(self, x_pad: torch.Tensor, t: int, dvec: Optional[torch.Tensor] = None,
                       lane: int = 0, x_amax: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Padded-layout entry used by the fused wrapper: [N,C,ldt] -> mask logits [N,C,ldt].
        `lane` selects the cached scratch buffer (one per concurrent HIP stream of the caller); `x_amax` [N, parts]:
        values whose per-utterance maximum bounds |x_pad| (only the fused normal-TCN path uses it)."""
        if self.tcn_layer.lower() == "gated":
            return self._forward_gated(x_pad, t, dvec)
        if not all(m.plan(x_pad.device)["fused"] for stack in self.tcn_list for m in stack):
            return self._forward_staged(x_pad, t, dvec)
        blocks, n_blocks = self.block_array(x_pad.device)
        need_embed = any(self.tcn_with_embed)
        if need_embed and dvec is None:
            # the reference would call TCN(x, None) on a block whose in_conv expects C+E channels and fail there
            raise RuntimeError("ConvTasNet.forward: tcn_with_embed is set but no dvec was given")
        if not need_embed:
            dvec = None  # reference ignores dvec when no block takes it (conv_tasnet.py:354-357)
        n = x_pad.shape[0]
        need = hip.lib().ps_conv_tasnet_workspace_bytes(n, self.input_dim, self.tcn_dim, t)
        if self._workspace is None:
            self._workspace = {}
        ws = self._workspace.get(lane)
        if ws is None or ws.numel() < need or ws.device != x_pad.device:
            ws = self._workspace[lane] = torch.zeros(need, dtype=torch.uint8, device=x_pad.device)
        rows_bf16 = all(m.plan(x_pad.device)["rows_bf16"] for stack in self.tcn_list for m in stack)
        return hip.conv_tasnet(blocks, n_blocks, x_pad, t, self.input_dim, self.tcn_dim,
                               None if dvec is None else dvec.contiguous().float(),
                               bool(self.embed_norm), ws, x_amax, bf16_rows=rows_bf16)

    def _forward_staged(self, x: torch.Tensor, t: int, dvec: Optional[torch.Tensor]) -> torch.Tensor:
        """Normal TCN blocks with a cLN somewhere: block by block, stage by stage."""
        if any(self.tcn_with_embed) and dvec is None:
            raise RuntimeError("ConvTasNet.forward: tcn_with_embed is set but no dvec was given")
        if dvec is not None and self.embed_norm:
            dvec = hip.l2_normalize(dvec.float())
        for stack in self.tcn_list:
            for i, blk in enumerate(stack):
                x = blk.forward_padded_staged(x, t, dvec if self.tcn_with_embed[i] else None)
        return x

    def _forward_gated(self, x: torch.Tensor, t: int, dvec: Optional[torch.Tensor]) -> torch.Tensor:
        """tcn_layer="gated": block by block (conv_tasnet.py:348-357)."""
        if dvec is not None and self.embed_norm:
            dvec = hip.l2_normalize(dvec.float())
        for stack in self.tcn_list:
            for i, blk in enumerate(stack):
                x = blk.forward_padded(x, t, dvec if (self.tcn_with_embed[i] and dvec is not None) else None)
        return x

    def forward(self, x: torch.Tensor, dvec: Optional[torch.Tensor] = None):
        """x [N,C,T], dvec [N,E] -> mask logits [N,C,T] (conv_tasnet.py:338-359)."""
        hip.require_device(x, "ConvTasNet.forward")
        t = x.shape[-1]
        return hip.unpad_rows(self.forward_padded(hip.pad_rows(x), t, dvec), t)

    @property
    def get_args(self) -> Dict:
        return {
            "input_dim": self.input_dim,
            "embed_dim": self.embed_dim,
            "embed_norm": self.embed_norm,
            "tcn_norm": self.tcn_norm,
            "dconv_norm": self.dconv_norm,
            "tcn_layer": self.tcn_layer,
            "tcn_dim": self.tcn_dim,
            "tcn_kernel": self.tcn_kernel,
            "tcn_dilated_basic": self.tcn_dilated_basic,
            "repeat_tcn": self.repeat_tcn,
            "per_tcn_stack": self.per_tcn_stack,
            "tcn_with_embed": self.tcn_with_embed,
            "causal": self.causal,
        }
