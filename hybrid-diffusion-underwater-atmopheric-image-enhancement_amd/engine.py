"""Host-side launch planner for the HIP kernels of the CFG-DDPM hot path.

PyTorch is used here only for device memory (``torch.empty``), streams and parameter storage.  All arithmetic is issued
through the C ABI of ``libhdiff.so`` (``include/hdiff.h``).  A :class:`Plan` is a static list of C-ABI launches over
preallocated buffers -- no allocation or synchronisation while it runs -- so a plan can be captured into a hipGraph
(``Plan.capture``) and replayed, which is how the T-step sampler loop runs (reference loop: DiffusionCondition.py:87-96).

The UNet emitter walks the architecture of the reference's ``UNet`` (ModelCondition.py:213-276) and maps it to launches:

  ResBlock (ModelCondition.py:196-211)          launches
    GroupNorm+Swish+Conv3x3 (+temb +cemb)   ->  gn_scale_shift (stats + fold, one launch), conv(prologue=GN/Swish, epilogue=+bias+vec);
                                                the temb / cemb projections of ALL blocks are one launch behind the embedding MLPs
    GroupNorm+Swish+Dropout+Conv3x3 + shortcut -> gn_scale_shift, [conv1x1 shortcut], conv(prologue, epilogue=+residual)
    MultiheadAttention                      ->  conv1x1 (packed in-proj) -> mha_flash_fwd -> conv1x1 (out-proj)
  DownSample (:74-76)  c1(x)+c2(x)          ->  ONE 5x5/s2 conv: the 3x3 weights are folded into the 5x5 centre at pack time
  UpSample (:85-89)    ConvTranspose 5x5/s2 ->  4 output-parity phases (3x3, 3x2, 2x3, 2x2 taps) as stride-1 convs, then conv3x3
  torch.cat skip (:271)                     ->  never materialised: consumers read two source pointers
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import _capi

GN_GROUPS = 32      # ModelCondition.py:170,184,249
GN_EPS = 1e-5
NUM_HEADS = 8       # ModelCondition.py:189


RESERVE_SPLIT_WORKSPACE = True      # plans reserve the split-operand attention scratch (Plan.attention_workspace); hdiff_amd.reserve_split_workspace


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _pad(v: int, m: int) -> int:
    return (v + m - 1) // m * m


def require_gpu_tensor(t: torch.Tensor, name: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"hdiff: '{name}' lives on {t.device}; the HIP path needs an MI355X device tensor "
                           "(there is deliberately no CPU fallback)")
    if t.dtype != torch.float32 and t.dtype != torch.int64 and t.dtype != torch.int32:
        raise RuntimeError(f"hdiff: '{name}' has dtype {t.dtype}; the path computes in fp32 with int64 indices")
    if not t.is_contiguous():
        raise RuntimeError(f"hdiff: '{name}' must be contiguous (NCHW)")


# ----------------------------------------------------------------------------------------------------------------------
# Packed convolution weights
# ----------------------------------------------------------------------------------------------------------------------
@dataclass
class TapSet:
    dy: List[int]
    dx: List[int]
    ky: List[int]      # kernel element feeding each tap (pack time)
    kx: List[int]


def conv_taps(k: int, pad: int) -> TapSet:
    ys = [ky for ky in range(k) for _ in range(k)]
    xs = [kx for _ in range(k) for kx in range(k)]
    return TapSet([y - pad for y in ys], [x - pad for x in xs], ys, xs)


def tconv_phase_taps(py: int, px: int) -> TapSet:
    """ConvTranspose2d(5, stride 2, pad 2, out_pad 1): output (2y+py, 2x+px) gathers x[y+dy][x+dx]*w[ky][kx] with
    ky = py + 2 - 2*dy (ModelCondition.py:83; oy = 2*iy - 2 + ky)."""
    kys = [ky for ky in range(5) if (ky - py) % 2 == 0]
    kxs = [kx for kx in range(5) if (kx - px) % 2 == 0]
    t = TapSet([], [], [], [])
    for ky in kys:
        for kx in kxs:
            t.dy.append((py + 2 - ky) // 2)
            t.dx.append((px + 2 - kx) // 2)
            t.ky.append(ky)
            t.kx.append(kx)
    return t


class PackedConv:
    """Device-side packed weights wp[tap][CinPad][CoutPad] for one convolution launch (conv_igemm.hip)."""

    def __init__(self, device, cout: int, cin: int, taps: TapSet):
        self.cout, self.cin, self.taps = cout, cin, taps
        self.cin_pad, self.cout_pad = _pad(cin, 8), _pad(cout, 64)
        self.ntaps = len(taps.dy)
        self.wp = torch.empty(self.ntaps * self.cin_pad * self.cout_pad, dtype=torch.float32, device=device)
        self.sources: List[Tuple[torch.Tensor, int, int, int, List[int], List[int], int]] = []
        self.wp3: Optional[torch.Tensor] = None      # split-bf16 copy (standard 3x3 convs with Cin % 16 == 0), see enable_x3
        self._x3_src: Optional[torch.Tensor] = None
        self._x3_transposed = False
        self._x3_taps: Optional[Tuple[int, int, int]] = None
        self.wp2: Optional[torch.Tensor] = None      # fp16-pair copy + the staged activations' power of two, see enable_h2
        self.act_scale: Optional[torch.Tensor] = None
        self._h2_src: Optional[torch.Tensor] = None
        self._h2_gn: Optional[Tuple[torch.Tensor, torch.Tensor, int, float]] = None

    def enable_h2(self, w: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, group_elems: int, gain: float = 1.0) -> None:
        """Also keep the weights as two fp16 pieces (conv3x3_x3.hip, PAIR) for a plain 3x3 conv whose input is the output of
        GroupNorm(gamma, beta) + Swish over groups of ``group_elems`` elements, times ``gain`` (1 / keep behind a dropout):
        that output is bounded by sqrt(n - 1) max|gamma| + max|beta|, which fixes the power of two the activations are staged
        with (hdiff_gn_act_scale, evaluated at pack time like the weights)."""
        ident = (gamma.data_ptr(), beta.data_ptr(), int(group_elems), float(gain))
        if self.wp2 is not None:
            # one pack = one staging scale: a second conv behind another GroupNorm (or another gain) must not inherit it silently
            have = (self._h2_gn[0].data_ptr(), self._h2_gn[1].data_ptr(), self._h2_gn[2], self._h2_gn[3])
            if have != ident:
                raise RuntimeError("PackedConv.enable_h2: this pack already stages activations for another GroupNorm / gain; "
                                   "use one PackedConv per (convolution, GroupNorm) pair")
            return
        if self.ntaps == 9 and self.cin % 16 == 0 and tuple(w.shape[2:]) == (3, 3):
            words = C.c_int64(0)
            _capi.check(_capi.lib().hdiff_pack_conv_weight_h2_words(self.cout, self.cin, self.cout_pad, C.byref(words)),
                        "pack_conv_weight_h2_words")
            self.wp2 = torch.empty(words.value, dtype=torch.int32, device=self.wp.device)
            self.act_scale = torch.empty(2, dtype=torch.float32, device=self.wp.device)
            self._h2_src = w
            self._h2_gn = (gamma, beta, int(group_elems), float(gain))

    def enable_x3(self, w: torch.Tensor, transposed: bool = False) -> None:
        """Also keep the weights as three bf16 pieces (conv3x3_x3.hip) -- used when the contraction mode is bf16x3.
        ``transposed``: ``w`` is the FORWARD conv's weight and this pack belongs to its input-gradient conv (same kernel on
        the transposed, tap-mirrored weight: cout / cin of this pack are the forward's cin / cout)."""
        if self.ntaps == 9 and self.cin % 16 == 0 and tuple(w.shape[2:]) == (3, 3):
            self.wp3 = torch.empty((self.cin // 16) * 9 * 3 * self.cout_pad * 8, dtype=torch.int32, device=self.wp.device)
            self._x3_src = w
            self._x3_transposed = bool(transposed)
            self._x3_taps = None

    def enable_x3_taps(self, w: torch.Tensor, mode: int) -> None:
        """The split-bf16 copy for a launch whose taps are a subset of the 3x3 neighbourhood reading elements (ky, kx) of a
        larger kernel: the output-parity phases of ConvTranspose2d(5, stride 2) (mode 1 = its [Cin][Cout][5][5] layout)."""
        if self.ntaps in (1, 4, 6, 9) and self.cin % 16 == 0 and all(-1 <= v <= 1 for v in self.taps.dy + self.taps.dx):
            self.wp3 = torch.empty((self.cin // 16) * self.ntaps * 3 * self.cout_pad * 8, dtype=torch.int32, device=self.wp.device)
            self._x3_src = w
            self._x3_taps = (int(mode), int(w.shape[2]), int(w.shape[3]))

    def add_source(self, w: torch.Tensor, mode: int, ky: Sequence[int], kx: Sequence[int], accumulate: int) -> None:
        kh, kw = int(w.shape[2]), int(w.shape[3])
        self.sources.append((w, mode, kh, kw, list(ky), list(kx), accumulate))

    def pack(self, stream: int) -> None:
        lib = _capi.lib()
        for w, mode, kh, kw, ky, kx, acc in self.sources:
            require_gpu_tensor(w, "conv weight")
            a_ky = (C.c_int * self.ntaps)(*ky)
            a_kx = (C.c_int * self.ntaps)(*kx)
            _capi.check(lib.hdiff_pack_conv_weight(w.data_ptr(), self.wp.data_ptr(), mode, self.cout, self.cin, kh, kw,
                                                   self.ntaps, a_ky, a_kx, self.cin_pad, self.cout_pad, acc, stream),
                        "pack_conv_weight")
        if self.wp3 is not None and self._x3_taps is not None:
            mode, kh, kw = self._x3_taps
            a_ky, a_kx = (C.c_int * self.ntaps)(*self.taps.ky), (C.c_int * self.ntaps)(*self.taps.kx)
            _capi.check(lib.hdiff_pack_conv_weight_x3_taps(self._x3_src.data_ptr(), self.wp3.data_ptr(), mode, self.cout, self.cin,
                                                           kh, kw, self.ntaps, a_ky, a_kx, self.cout_pad, stream),
                        "pack_conv_weight_x3_taps")
        elif self.wp3 is not None:
            _capi.check(lib.hdiff_pack_conv_weight_x3(self._x3_src.data_ptr(), self.wp3.data_ptr(), self.cout, self.cin,
                                                      self.cout_pad, int(self._x3_transposed), stream), "pack_conv_weight_x3")
        if self.wp2 is not None:
            gamma, beta, group_elems, gain = self._h2_gn
            _capi.check(lib.hdiff_pack_conv_weight_h2(self._h2_src.data_ptr(), self.wp2.data_ptr(), self.cout, self.cin,
                                                      self.cout_pad, stream), "pack_conv_weight_h2")
            _capi.check(lib.hdiff_gn_act_scale(gamma.data_ptr(), beta.data_ptr(), int(gamma.numel()), C.c_int64(group_elems),
                                               C.c_float(gain), self.act_scale.data_ptr(), stream), "gn_act_scale")


# ----------------------------------------------------------------------------------------------------------------------
# Plan
# ----------------------------------------------------------------------------------------------------------------------
class Plan:
    """A static, replayable list of C-ABI launches over preallocated device buffers."""

    def __init__(self, device: torch.device):
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("hdiff: plans run on an MI355X device only (no CPU fallback)")
        self.lib = _capi.lib()
        self.ops: List[Tuple[str, Callable, tuple]] = []
        self.packs: List[PackedConv] = []
        self._gn_src: Dict[int, Tuple[torch.Tensor, torch.Tensor, int, float]] = {}   # id(gn scale buffer) -> its GroupNorm
        self._pool: Dict[int, List[torch.Tensor]] = {}
        self._all: List[torch.Tensor] = []
        self._keep: List[object] = []
        self.graph = C.c_void_p(None)
        self._graph_stream: Optional[torch.cuda.Stream] = None
        self.flops = 0.0          # algorithmic FLOPs of the MFMA-bound launches (convolutions, attention) of one run
        self._vec_jobs: List[tuple] = []          # per-block embedding projections, emitted as ONE launch (flush_block_vecs)

    # -- memory -------------------------------------------------------------------------------------------------------
    def buf(self, *shape: int, dtype=torch.float32) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= int(s)
        key = n * (8 if dtype == torch.int64 else 4)
        free = self._pool.get(key)
        if free:
            base = free.pop()
        else:
            base = torch.empty(key // 4, dtype=torch.float32, device=self.device)
            self._all.append(base)
        t = base.view(dtype)[:n].view(*shape) if dtype != torch.float32 else base[:n].view(*shape)
        t._hdiff_base = base  # type: ignore[attr-defined]
        return t

    def free(self, t: torch.Tensor) -> None:
        base = getattr(t, "_hdiff_base", None)
        if base is not None:
            self._pool.setdefault(base.numel() * 4, []).append(base)

    def bytes_allocated(self) -> int:
        return sum(b.numel() * 4 for b in self._all)

    # -- launches -----------------------------------------------------------------------------------------------------
    def call(self, name: str, *args) -> None:
        self.ops.append((name, getattr(self.lib, name), args))

    def keep(self, obj) -> None:
        self._keep.append(obj)

    def run(self, stream: Optional[int] = None) -> None:
        # launches go to the plan's device whatever the caller's current device is (the C ABI launches on the current
        # device; per-kernel attributes such as the dynamic-LDS limit are set per device on first use)
        assert not self._vec_jobs, "flush_block_vecs() was not called for this plan"
        with torch.cuda.device(self.device):
            s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
            for name, fn, args in self.ops:
                rc = fn(*args, s)
                if rc != 0:
                    _capi.check(rc, name)

    def pack_weights(self, stream: Optional[int] = None) -> None:
        with torch.cuda.device(self.device):
            s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
            for p in self.packs:
                p.pack(s)

    # -- hipGraph -----------------------------------------------------------------------------------------------------
    def capture(self) -> None:
        """Capture the launch list into a hipGraph on a private stream (weights must already be packed)."""
        if self.graph.value:
            return
        with torch.cuda.device(self.device):
            torch.cuda.synchronize(self.device)
            self._graph_stream = torch.cuda.Stream(self.device)
            s = self._graph_stream.cuda_stream
            _capi.check(self.lib.hdiff_graph_begin(s), "graph_begin")
            try:
                self.run(s)
            finally:
                rc = self.lib.hdiff_graph_end(s, C.byref(self.graph))
            _capi.check(rc, "graph_end")

    def replay(self, stream: Optional[int] = None) -> None:
        with torch.cuda.device(self.device):
            s = torch.cuda.current_stream(self.device).cuda_stream if stream is None else stream
            _capi.check(self.lib.hdiff_graph_launch(self.graph, s), "graph_launch")

    def __del__(self):
        try:
            if self.graph.value:
                self.lib.hdiff_graph_destroy(self.graph)
        except Exception:
            pass

    # -- op emitters --------------------------------------------------------------------------------------------------
    def conv(self, x0: torch.Tensor, x1: Optional[torch.Tensor], pk: PackedConv, bias: Optional[torch.Tensor],
             out: torch.Tensor, *, B: int, H: int, W: int, VH: int, VW: int, in_stride: int = 1,
             out_map: Tuple[int, int, int, int] = (1, 0, 1, 0), gn: Optional[Tuple[torch.Tensor, torch.Tensor]] = None,
             addvec: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None,
             act_range: Optional[Tuple[torch.Tensor, torch.Tensor, int, float]] = None) -> None:
        """act_range = (gamma, beta, group_elems, gain): the input tensor itself is gain * swish(GroupNorm(gamma, beta)(.)) of
        groups of group_elems elements (an activation materialised by the caller, e.g. behind a dropout mask) -- what the
        fp16-pair 3x3 kernel needs to know about its range; with ``gn`` from gn_scale_shift() the plan knows it already."""
        d = _capi.ConvDesc()
        C0 = int(x0.shape[1])
        C1 = int(x1.shape[1]) if x1 is not None else 0
        assert C0 + C1 == pk.cin, (C0, C1, pk.cin)
        d.x0, d.x1, d.C0, d.C1 = _ptr(x0), _ptr(x1), C0, C1
        d.B, d.H, d.W = B, H, W
        d.wp, d.bias = pk.wp.data_ptr(), _ptr(bias)
        d.Cout, d.CinPad, d.CoutPad = pk.cout, pk.cin_pad, pk.cout_pad
        d.gn_scale, d.gn_shift = (_ptr(gn[0]), _ptr(gn[1])) if gn is not None else (None, None)
        d.addvec, d.residual, d.out = _ptr(addvec), _ptr(residual), out.data_ptr()
        d.OH, d.OW = int(out.shape[2]), int(out.shape[3])
        d.VH, d.VW, d.in_stride = VH, VW, in_stride
        d.out_sy, d.out_oy, d.out_sx, d.out_ox = out_map
        d.ntaps = pk.ntaps
        d.wp_x3 = _ptr(pk.wp3)
        if act_range is None and gn is not None:
            act_range = self._gn_src.get(id(gn[0]))
        if act_range is not None and pk.wp3 is not None and pk._x3_taps is None and not pk._x3_transposed and in_stride == 1 \
                and out_map == (1, 0, 1, 0):
            # a plain 3x3 conv behind GroupNorm + Swish: the fp16-pair form of the split-operand kernel (its input range is known)
            pk.enable_h2(pk._x3_src, *act_range)
        if act_range is not None:      # only calls that carry the range take the fp16-pair form (a pack reused without one runs the triples)
            d.wp_h2, d.act_scale = _ptr(pk.wp2), _ptr(pk.act_scale)
        for i in range(pk.ntaps):
            d.tap_dy[i], d.tap_dx[i] = pk.taps.dy[i], pk.taps.dx[i]
        need = C.c_int64(0)
        _capi.check(self.lib.hdiff_conv2d_fwd_workspace(C.byref(d), C.byref(need)), "conv2d_fwd_workspace")
        ws = None
        if need.value > 0:               # small grid, long channel loop: split-K partial sums + ordered reduce
            ws = self.buf(need.value)
            d.splitk_ws, d.splitk_floats = ws.data_ptr(), need.value
        self.keep((d, x0, x1, pk, bias, out, gn, addvec, residual, ws))
        self.flops += 2.0 * pk.ntaps * pk.cin * pk.cout * VH * VW * B
        self.call("hdiff_conv2d_fwd", C.byref(d))
        if ws is not None:
            self.free(ws)

    def attention_workspace(self, B: int, Cc: int, heads: int, L: int) -> Optional[torch.Tensor]:
        """Scratch for hdiff_mha_flash_fwd_ws (the operands as fp16 pieces, written and read inside that one call when the
        contraction mode is bf16x3): sized by the library from the SHAPE alone and allocated whatever the mode is now -- plans are
        keyed by shape, not by mode, and a plan first built in the f32 mode must not run the slower in-loop-split kernel once the
        mode is switched (tests/test_gpu_end_to_end.py builds its plan in f32 and asserts the pre-split kernel ran).
        COST: 18 bytes per qkv element plus a small tail -- ~2.4 GB at B = 16, C = 128, L = 65 536 -- that a plan which only
        ever runs in the f32 mode never touches.  A deployment that stays in f32 opts out BEFORE it builds its plans with
        ``hdiff_amd.reserve_split_workspace(False)``; a plan built that way, run in the split-operand mode after all, falls back to the
        in-loop-split kernel (correct, slower) -- the library decides per call from the pointer it is given."""
        if not RESERVE_SPLIT_WORKSPACE:      # hdiff_amd.reserve_split_workspace(False): an f32-only deployment keeps the memory
            return None
        need = C.c_int64(0)
        _capi.check(self.lib.hdiff_mha_flash_fwd_workspace(B, Cc, heads, L, C.byref(need)), "mha_flash_fwd_workspace")
        return self.buf((need.value + 3) // 4) if need.value > 0 else None

    def gn_scale_shift(self, x0: torch.Tensor, x1: Optional[torch.Tensor], gamma: torch.Tensor, beta: torch.Tensor,
                       B: int, HW: int) -> Tuple[torch.Tensor, torch.Tensor]:
        C0 = int(x0.shape[1])
        C1 = int(x1.shape[1]) if x1 is not None else 0
        Ct = C0 + C1
        if Ct % GN_GROUPS != 0:
            raise RuntimeError(f"Expected number of channels in input to be divisible by num_groups, got {Ct}")
        # enough workgroups to stream at HBM rate, at least 4K elements per slice
        nsplit = max(1, min(32, HW // 4096))     # a function of the plane size only: per-sample results must not depend on B
        ws = self.buf(B * GN_GROUPS * nsplit * 3)
        scale, shift = self.buf(B, Ct), self.buf(B, Ct)
        # statistics and the (mean, rstd, gamma, beta) -> scale / shift fold behind one entry point: one launch when a
        # (sample, group) is a single workgroup (planes up to 64x64), the streaming pass + the small merge kernel otherwise
        self.call("hdiff_gn_scale_shift", _ptr(x0), _ptr(x1), C0, C1, B, HW, GN_GROUPS, nsplit, ws.data_ptr(),
                  gamma.data_ptr(), beta.data_ptr(), C.c_float(GN_EPS), scale.data_ptr(), shift.data_ptr())
        self.keep((x0, x1, gamma, beta, ws))
        self.free(ws)
        self._gn_src[id(scale)] = (gamma, beta, (Ct // GN_GROUPS) * HW, 1.0)     # for conv(): the range of swish(gn(x))
        return scale, shift

    def block_vec(self, temb: torch.Tensor, cemb: Optional[torch.Tensor], P: Dict[str, torch.Tensor], p: str, B: int,
                  cout: int) -> torch.Tensor:
        """temb_proj(temb) + cond_proj(cemb) of one ResBlock (ModelCondition.py:199-200) as a [B][cout] vector for the conv
        epilogue.  All of a plan's projections depend only on temb / cemb, so they are collected here and emitted as ONE
        launch at the point `flush_block_vecs` is told (right behind the embedding MLPs)."""
        vec = torch.empty(B, cout, dtype=torch.float32, device=self.device)       # lives as long as the plan: not pooled
        wc = P[f"{p}.cond_proj.1.weight"] if cemb is not None else None
        bc = P[f"{p}.cond_proj.1.bias"] if cemb is not None else None
        self._vec_jobs.append((P[f"{p}.temb_proj.1.weight"], P[f"{p}.temb_proj.1.bias"], wc, bc, vec))
        return vec

    def flush_block_vecs(self, pos: int, temb: torch.Tensor, cemb: Optional[torch.Tensor], B: int) -> None:
        """Insert the one hdiff_linear_rows_multi launch for every collected projection at op index `pos`."""
        if not self._vec_jobs:
            return
        jobs = (_capi.LinearJob * len(self._vec_jobs))()
        first = 0
        for j, (wt, bt, wc, bc, vec) in enumerate(self._vec_jobs):
            jobs[j].w0, jobs[j].b0, jobs[j].w1, jobs[j].b1 = wt.data_ptr(), bt.data_ptr(), _ptr(wc), _ptr(bc)
            jobs[j].y, jobs[j].n, jobs[j].first = vec.data_ptr(), int(wt.shape[0]), first
            first += int(wt.shape[0])
        raw = bytes(jobs)
        table = torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(self.device)
        K = int(self._vec_jobs[0][0].shape[1])
        # ONE input width serves every job and both inputs (the kernel reads K floats of temb and of cemb per row)
        if int(temb.shape[1]) != K or (cemb is not None and int(cemb.shape[1]) != K):
            raise RuntimeError(f"block projections: embedding width {tuple(temb.shape)} / "
                               f"{None if cemb is None else tuple(cemb.shape)} does not match the weights' {K}")
        for wt, _, wc, _, _ in self._vec_jobs:
            if int(wt.shape[1]) != K or (wc is not None and int(wc.shape[1]) != K):
                raise RuntimeError("block projections: temb_proj / cond_proj weights of different input widths in one model")
        self.keep((table, temb, cemb, list(self._vec_jobs)))
        self.ops.insert(pos, ("hdiff_linear_rows_multi", getattr(self.lib, "hdiff_linear_rows_multi"),
                              (temb.data_ptr(), _ptr(cemb), table.data_ptr(), len(self._vec_jobs), first, B, K)))
        self._vec_jobs = []

    def linear(self, x: torch.Tensor, idx: Optional[torch.Tensor], W: torch.Tensor, bias: Optional[torch.Tensor],
               y: torch.Tensor, B: int, swish_input: bool, accumulate: bool) -> None:
        N, K = int(W.shape[0]), int(W.shape[1])
        self.keep((x, idx, W, bias, y))
        n_rows = int(x.shape[0]) if idx is not None else 0
        self.call("hdiff_linear_rows", x.data_ptr(), _ptr(idx), n_rows, W.data_ptr(), _ptr(bias), y.data_ptr(), B, K, N,
                  int(swish_input), int(accumulate))


# ----------------------------------------------------------------------------------------------------------------------
# UNet emitter
# ----------------------------------------------------------------------------------------------------------------------
@dataclass
class UNetShape:
    T: int
    num_labels: int
    ch: int
    ch_mult: Tuple[int, ...]
    num_res_blocks: int


def _new_pack(plan: Plan, cout: int, cin: int, taps: TapSet) -> PackedConv:
    pk = PackedConv(plan.device, cout, cin, taps)
    plan.packs.append(pk)
    return pk


def _std_pack(plan: Plan, w: torch.Tensor, k: int, pad: int) -> PackedConv:
    taps = conv_taps(k, pad)
    pk = _new_pack(plan, int(w.shape[0]), int(w.shape[1]), taps)
    pk.add_source(w, 0, taps.ky, taps.kx, 0)
    if k == 3 and pad == 1:
        pk.enable_x3(w)
    elif k == 1 and pad == 0:
        pk.enable_x3_taps(w, 0)          # one tap: the 1x1 GEMM on bf16 triples (conv1x1_x3.hip) in the bf16x3 mode
    return pk


def emit_embed_mlp(plan: Plan, P: Dict[str, torch.Tensor], prefix: str, idx: torch.Tensor, B: int) -> torch.Tensor:
    """Embedding -> Linear -> Swish -> Linear (ModelCondition.py:38-49, 56-65)."""
    w1, b1, w2, b2 = P[f"{prefix}.1.weight"], P[f"{prefix}.1.bias"], P[f"{prefix}.3.weight"], P[f"{prefix}.3.bias"]
    h = plan.buf(B, int(w1.shape[0]))
    out = plan.buf(B, int(w2.shape[0]))
    plan.linear(P[f"{prefix}.0.weight"], idx, w1, b1, h, B, swish_input=False, accumulate=False)
    plan.linear(h, None, w2, b2, out, B, swish_input=True, accumulate=False)
    plan.free(h)
    return out


def emit_mha(plan: Plan, P: Dict[str, torch.Tensor], p: str, h: torch.Tensor, B: int, Cc: int, H: int, W: int) -> torch.Tensor:
    """nn.MultiheadAttention(C, 8) as attn(h,h,h) on (L,B,C): no pre-norm, no residual (ModelCondition.py:204-208)."""
    if Cc % NUM_HEADS != 0:
        raise AssertionError("embed_dim must be divisible by num_heads")
    w_in = P[f"{p}.attn.in_proj_weight"].view(3 * Cc, Cc, 1, 1)
    w_out = P[f"{p}.attn.out_proj.weight"].view(Cc, Cc, 1, 1)
    pk_in, pk_out = _std_pack(plan, w_in, 1, 0), _std_pack(plan, w_out, 1, 0)
    qkv = plan.buf(B, 3 * Cc, H, W)
    plan.conv(h, None, pk_in, P[f"{p}.attn.in_proj_bias"], qkv, B=B, H=H, W=W, VH=H, VW=W)
    o = plan.buf(B, Cc, H, W)
    ws = plan.attention_workspace(B, Cc, NUM_HEADS, H * W)
    plan.call("hdiff_mha_flash_fwd_ws", qkv.data_ptr(), o.data_ptr(), None, B, Cc, NUM_HEADS, H * W, _ptr(ws),
              C.c_int64(0 if ws is None else ws.numel() * 4))
    plan.flops += 4.0 * (H * W) ** 2 * Cc * B
    plan.keep((qkv, o, ws))
    plan.free(qkv)
    if ws is not None:
        plan.free(ws)
    y = plan.buf(B, Cc, H, W)
    plan.conv(o, None, pk_out, P[f"{p}.attn.out_proj.bias"], y, B=B, H=H, W=W, VH=H, VW=W)
    plan.free(o)
    return y


def emit_attn_block(plan: Plan, P: Dict[str, torch.Tensor], p: str, x: torch.Tensor, B: int, Cc: int, H: int, W: int) -> torch.Tensor:
    """AttnBlock.forward (ModelCondition.py:102-120): GroupNorm (no Swish) -> q, k, v as ONE stacked 1x1 conv -> single-head
    attention of width C (scale C^-1/2) -> 1x1 proj with the residual x fused into its epilogue."""
    L = H * W
    scale, shift = plan.gn_scale_shift(x, None, P[f"{p}.group_norm.weight"], P[f"{p}.group_norm.bias"], B, L)
    hn = plan.buf(B, Cc, H, W)
    plan.call("hdiff_gn_affine_apply", x.data_ptr(), scale.data_ptr(), shift.data_ptr(), hn.data_ptr(), B, Cc, L)
    plan.keep((x, scale, shift, hn))
    plan.free(scale); plan.free(shift)
    # rows [Wq | Wk | Wv]: the layout the attention kernels read (same as nn.MultiheadAttention's packed in-projection)
    w_qkv = torch.cat([P[f"{p}.proj_q.weight"], P[f"{p}.proj_k.weight"], P[f"{p}.proj_v.weight"]], dim=0).detach().contiguous()
    b_qkv = torch.cat([P[f"{p}.proj_q.bias"], P[f"{p}.proj_k.bias"], P[f"{p}.proj_v.bias"]], dim=0).detach().contiguous()
    plan.keep((w_qkv, b_qkv))
    pk_in = _std_pack(plan, w_qkv, 1, 0)
    qkv = plan.buf(B, 3 * Cc, H, W)
    plan.conv(hn, None, pk_in, b_qkv, qkv, B=B, H=H, W=W, VH=H, VW=W)
    plan.free(hn)
    o = plan.buf(B, Cc, H, W)
    if Cc <= 64:
        ws = plan.attention_workspace(B, Cc, 1, L)
        plan.call("hdiff_mha_flash_fwd_ws", qkv.data_ptr(), o.data_ptr(), None, B, Cc, 1, L, _ptr(ws),     # one head of width C
                  C.c_int64(0 if ws is None else ws.numel() * 4))
        plan.keep(ws)
        if ws is not None:
            plan.free(ws)
    else:
        plan.call("hdiff_mha_wide_fwd", qkv.data_ptr(), o.data_ptr(), B, Cc, L)
    plan.flops += 4.0 * L * L * Cc * B
    plan.keep((qkv, o))
    plan.free(qkv)
    pk_out = _std_pack(plan, P[f"{p}.proj.weight"], 1, 0)
    y = plan.buf(B, Cc, H, W)
    plan.conv(o, None, pk_out, P[f"{p}.proj.bias"], y, B=B, H=H, W=W, VH=H, VW=W, residual=x)
    plan.free(o)
    return y


def emit_resblock(plan: Plan, P: Dict[str, torch.Tensor], p: str, xa: torch.Tensor, xb: Optional[torch.Tensor],
                  temb: torch.Tensor, cemb: Optional[torch.Tensor], cout: int, B: int, H: int, W: int, attn: bool) -> torch.Tensor:
    """ResBlock.forward (ModelCondition.py:196-211) on the virtual concat [xa | xb]."""
    sc1 = plan.gn_scale_shift(xa, xb, P[f"{p}.block1.0.weight"], P[f"{p}.block1.0.bias"], B, H * W)
    vec = plan.block_vec(temb, cemb, P, p, B, cout)
    pk1 = _std_pack(plan, P[f"{p}.block1.2.weight"], 3, 1)
    h1 = plan.buf(B, cout, H, W)
    plan.conv(xa, xb, pk1, P[f"{p}.block1.2.bias"], h1, B=B, H=H, W=W, VH=H, VW=W, gn=sc1, addvec=vec)
    plan.free(sc1[0]); plan.free(sc1[1])

    sc2 = plan.gn_scale_shift(h1, None, P[f"{p}.block2.0.weight"], P[f"{p}.block2.0.bias"], B, H * W)
    if f"{p}.shortcut.weight" in P:
        pks = _std_pack(plan, P[f"{p}.shortcut.weight"], 1, 0)
        res = plan.buf(B, cout, H, W)
        plan.conv(xa, xb, pks, P[f"{p}.shortcut.bias"], res, B=B, H=H, W=W, VH=H, VW=W)
        res_owned = True
    else:
        assert xb is None
        res, res_owned = xa, False
    pk2 = _std_pack(plan, P[f"{p}.block2.3.weight"], 3, 1)
    h2 = plan.buf(B, cout, H, W)
    plan.conv(h1, None, pk2, P[f"{p}.block2.3.bias"], h2, B=B, H=H, W=W, VH=H, VW=W, gn=sc2, residual=res)
    plan.free(sc2[0]); plan.free(sc2[1]); plan.free(h1)
    if res_owned:
        plan.free(res)
    if attn:
        y = emit_mha(plan, P, p, h2, B, cout, H, W)
        plan.free(h2)
        return y
    return h2


def emit_downsample(plan: Plan, P: Dict[str, torch.Tensor], p: str, x: torch.Tensor, B: int, Cc: int, H: int, W: int) -> torch.Tensor:
    """DownSample.forward (ModelCondition.py:74-76): c1(x) + c2(x) as one 5x5/s2 conv with folded weights and biases."""
    taps = conv_taps(5, 2)
    pk = _new_pack(plan, Cc, Cc, taps)
    pk.add_source(P[f"{p}.c2.weight"], 0, taps.ky, taps.kx, 0)
    ky3 = [ky - 1 if (1 <= ky <= 3 and 1 <= kx <= 3) else -1 for ky, kx in zip(taps.ky, taps.kx)]
    kx3 = [kx - 1 if (1 <= ky <= 3 and 1 <= kx <= 3) else 0 for ky, kx in zip(taps.ky, taps.kx)]
    pk.add_source(P[f"{p}.c1.weight"], 0, ky3, kx3, 1)
    OH, OW = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    bias = plan.buf(Cc)
    plan.call("hdiff_axpby", C.c_float(1.0), P[f"{p}.c1.bias"].data_ptr(), C.c_float(1.0), P[f"{p}.c2.bias"].data_ptr(),
              bias.data_ptr(), Cc)
    plan.keep((P[f"{p}.c1.bias"], P[f"{p}.c2.bias"]))
    y = plan.buf(B, Cc, OH, OW)
    plan.conv(x, None, pk, bias, y, B=B, H=H, W=W, VH=OH, VW=OW, in_stride=2)
    plan.free(bias)
    return y


def emit_upsample(plan: Plan, P: Dict[str, torch.Tensor], p: str, x: torch.Tensor, B: int, Cc: int, H: int, W: int) -> torch.Tensor:
    """UpSample.forward (ModelCondition.py:85-89): ConvTranspose2d(5,2,2,1) as 4 parity phases, then Conv3x3."""
    wt = P[f"{p}.t.weight"]                      # [Cin][Cout][5][5]
    u = plan.buf(B, Cc, 2 * H, 2 * W)
    for py in (0, 1):
        for px in (0, 1):
            taps = tconv_phase_taps(py, px)
            pk = _new_pack(plan, Cc, Cc, taps)
            pk.add_source(wt, 1, taps.ky, taps.kx, 0)
            pk.enable_x3_taps(wt, 1)             # every phase's taps lie in the 3x3 neighbourhood: the split-bf16 kernel serves them
            plan.conv(x, None, pk, P[f"{p}.t.bias"], u, B=B, H=H, W=W, VH=H, VW=W, out_map=(2, py, 2, px))
    pkc = _std_pack(plan, P[f"{p}.c.weight"], 3, 1)
    y = plan.buf(B, Cc, 2 * H, 2 * W)
    plan.conv(u, None, pkc, P[f"{p}.c.bias"], y, B=B, H=2 * H, W=2 * W, VH=2 * H, VW=2 * W)
    plan.free(u)
    return y


class UNetPlan:
    """Launch plan of one UNet.forward (ModelCondition.py:255-276) for a fixed (B, H, W)."""

    def __init__(self, P: Dict[str, torch.Tensor], shape: UNetShape, B: int, H: int, W: int, device):
        if H % (2 ** (len(shape.ch_mult) - 1)) or W % (2 ** (len(shape.ch_mult) - 1)):
            # the reference fails at the skip concat (ModelCondition.py:271) for such sizes
            raise RuntimeError(f"Sizes of tensors must match: H, W = {H}, {W} not divisible by "
                               f"{2 ** (len(shape.ch_mult) - 1)}")
        plan = Plan(device)
        self.plan, self.B, self.H, self.W = plan, B, H, W
        self.x = plan.buf(B, 3, H, W)
        self.t = plan.buf(B, dtype=torch.int64)
        self.labels = plan.buf(B, dtype=torch.int64)
        ch = shape.ch
        temb = emit_embed_mlp(plan, P, "time_embedding.timembedding", self.t, B)
        cemb = emit_embed_mlp(plan, P, "cond_embedding.condEmbedding", self.labels, B)
        vec_pos = len(plan.ops)            # the per-block projections of temb / cemb go here, as one launch

        pk_head = _std_pack(plan, P["head.weight"], 3, 1)
        h = plan.buf(B, ch, H, W)
        plan.conv(self.x, None, pk_head, P["head.bias"], h, B=B, H=H, W=W, VH=H, VW=W)
        hs: List[Tuple[torch.Tensor, int, int, int]] = [(h, ch, H, W)]
        now, cH, cW = ch, H, W
        n_down = 0
        for i, mult in enumerate(shape.ch_mult):
            out = ch * mult
            for _ in range(shape.num_res_blocks):
                h = emit_resblock(plan, P, f"downblocks.{n_down}", h, None, temb, cemb, out, B, cH, cW, attn=True)
                n_down += 1
                now = out
                hs.append((h, now, cH, cW))
            if i != len(shape.ch_mult) - 1:
                h = emit_downsample(plan, P, f"downblocks.{n_down}", h, B, now, cH, cW)
                n_down += 1
                cH, cW = (cH - 1) // 2 + 1, (cW - 1) // 2 + 1
                hs.append((h, now, cH, cW))
        hm = emit_resblock(plan, P, "middleblocks.0", h, None, temb, cemb, now, B, cH, cW, attn=True)
        h = emit_resblock(plan, P, "middleblocks.1", hm, None, temb, cemb, now, B, cH, cW, attn=False)
        plan.free(hm)
        n_up = 0
        for i, mult in reversed(list(enumerate(shape.ch_mult))):
            out = ch * mult
            for _ in range(shape.num_res_blocks + 1):
                skip, sc, sH, sW = hs.pop()
                assert (sH, sW) == (cH, cW)
                y = emit_resblock(plan, P, f"upblocks.{n_up}", h, skip, temb, cemb, out, B, cH, cW, attn=False)
                n_up += 1
                plan.free(h); plan.free(skip)
                h, now = y, out
            if i != 0:
                y = emit_upsample(plan, P, f"upblocks.{n_up}", h, B, now, cH, cW)
                n_up += 1
                plan.free(h)
                h, cH, cW = y, 2 * cH, 2 * cW
        assert not hs
        sct = plan.gn_scale_shift(h, None, P["tail.0.weight"], P["tail.0.bias"], B, cH * cW)
        pk_tail = _std_pack(plan, P["tail.2.weight"], 3, 1)
        self.out = plan.buf(B, 3, H, W)
        plan.conv(h, None, pk_tail, P["tail.2.bias"], self.out, B=B, H=H, W=W, VH=H, VW=W, gn=sct)
        plan.flush_block_vecs(vec_pos, temb, cemb, B)
        plan.keep((temb, cemb, h))


# ----------------------------------------------------------------------------------------------------------------------
# DynamicUNet emitter (the reference's second tree: diffusion/Model.py:382-517)
# ----------------------------------------------------------------------------------------------------------------------
@dataclass
class DynUNetShape:
    T: int
    ch: int
    ch_mult: Tuple[int, ...]
    num_res_blocks: int


def emit_cond_image_embedding(plan: Plan, P: Dict[str, torch.Tensor], p: str, img: torch.Tensor, B: int, H: int, W: int) -> torch.Tensor:
    """ConditionalEmbedding.forward (diffusion/Model.py:135-166): three stride-2 3x3 convs with no activation in between,
    global average pool, Linear -> Swish -> Linear."""
    x, cH, cW = img, H, W
    for name in ("conv1", "conv2", "conv3"):
        w = P[f"{p}.{name}.weight"]
        pk = _std_pack(plan, w, 3, 1)
        oH, oW = (cH - 1) // 2 + 1, (cW - 1) // 2 + 1
        y = plan.buf(B, int(w.shape[0]), oH, oW)
        plan.conv(x, None, pk, P[f"{p}.{name}.bias"], y, B=B, H=cH, W=cW, VH=oH, VW=oW, in_stride=2)
        if x is not img:
            plan.free(x)
        x, cH, cW = y, oH, oW
    Cc = int(x.shape[1])
    pooled = plan.buf(B, Cc)
    plan.call("hdiff_avgpool_global", x.data_ptr(), pooled.data_ptr(), B * Cc, cH * cW)
    plan.keep((x, pooled))
    plan.free(x)
    w1, w2 = P[f"{p}.linear1.weight"], P[f"{p}.linear2.weight"]
    h = plan.buf(B, int(w1.shape[0]))
    out = plan.buf(B, int(w2.shape[0]))
    plan.linear(pooled, None, w1, P[f"{p}.linear1.bias"], h, B, swish_input=False, accumulate=False)
    plan.linear(h, None, w2, P[f"{p}.linear2.bias"], out, B, swish_input=True, accumulate=False)
    plan.free(pooled); plan.free(h)
    return out


class DynUNetPlan:
    """Launch plan of one DynamicUNet.forward (diffusion/Model.py:475-515) for a fixed (B, H, W).

    The inputs are two 3-channel tensors (``cond`` = the conditioning image, ``y`` = the noisy image, which the sampler
    updates in place); their ``torch.cat`` (Diffusion.py:229,252) is one small launch into ``x6`` -- 3 channels are too
    narrow for the conv's 4-channel-aligned two-pointer input.  ``context_zero`` selects a zero conditional embedding
    (what the reference's sampler always uses) or the conv embedding of ``label`` (an image)."""

    def __init__(self, P: Dict[str, torch.Tensor], shape: DynUNetShape, B: int, H: int, W: int, device, context_zero: bool,
                 taps: Optional[Dict[str, torch.Tensor]] = None):
        plan = Plan(device)
        if taps is not None:               # test hook: keep every block output alive (no buffer reuse) and name it
            plan.free = lambda t: None     # type: ignore[method-assign]
        else:
            taps = {}
        self.plan, self.B, self.H, self.W = plan, B, H, W
        self.cond = plan.buf(B, 3, H, W)
        self.y = plan.buf(B, 3, H, W)
        self.t = plan.buf(B, dtype=torch.int64)
        self.label = None if context_zero else plan.buf(B, 3, H, W)
        ch = shape.ch
        temb = emit_embed_mlp(plan, P, "time_embedding.timembedding", self.t, B)
        if context_zero:
            # torch.zeros_like(temb), Model.py:482-483.  Its own allocation, NOT a pooled buffer: it is written once here, and a
            # pooled buffer would be the recycled scratch of the time MLP, rewritten on every run
            cemb = torch.zeros(B, 4 * ch, dtype=torch.float32, device=plan.device)
            self._zero = cemb
        else:
            cemb = emit_cond_image_embedding(plan, P, "cond_embedding", self.label, B, H, W)
            self._zero = None

        vec_pos = len(plan.ops)            # the per-block projections of temb / cemb go here, as one launch
        self.x6 = plan.buf(B, 6, H, W)
        plan.call("hdiff_concat2", self.cond.data_ptr(), self.y.data_ptr(), self.x6.data_ptr(), B, 3 * H * W, 3 * H * W)
        pk_head = _std_pack(plan, P["head.weight"], 3, 1)
        h = plan.buf(B, ch, H, W)
        plan.conv(self.x6, None, pk_head, P["head.bias"], h, B=B, H=H, W=W, VH=H, VW=W)
        taps["temb"], taps["cemb"], taps["head"] = temb, cemb, h
        hs: List[Tuple[torch.Tensor, int, int, int]] = [(h, ch, H, W)]
        now, cH, cW = ch, H, W
        n = 0
        for i, mult in enumerate(shape.ch_mult):
            for _ in range(shape.num_res_blocks):
                h = emit_resblock(plan, P, f"downblocks.{n}", h, None, temb, cemb, ch * mult, B, cH, cW, attn=False)
                taps[f"downblocks.{n}"] = h
                n += 1
                now = ch * mult
                hs.append((h, now, cH, cW))
            if i != len(shape.ch_mult) - 1:
                h = emit_downsample(plan, P, f"downblocks.{n}", h, B, now, cH, cW)
                taps[f"downblocks.{n}"] = h
                n += 1
                cH, cW = (cH - 1) // 2 + 1, (cW - 1) // 2 + 1
                hs.append((h, now, cH, cW))
        owned = False                      # the last down output is also a skip tensor: it must outlive the middle
        for k in range(4):                 # four attention ResBlocks (Model.py:425-431)
            y = emit_resblock(plan, P, f"middleblocks.{k}", h, None, temb, cemb, now, B, cH, cW, attn=True)
            if owned:
                plan.free(h)
            h, owned = y, True
            taps[f"middleblocks.{k}"] = h
        n = 0
        for i, mult in reversed(list(enumerate(shape.ch_mult))):
            for _ in range(shape.num_res_blocks):
                skip, sc, sH, sW = hs.pop()
                if (sH, sW) != (cH, cW):   # F.interpolate(skip, size=h.shape[2:], mode="nearest"), Model.py:503-504
                    r = plan.buf(B, sc, cH, cW)
                    plan.call("hdiff_resize_nearest", skip.data_ptr(), r.data_ptr(), B * sc, sH, sW, cH, cW)
                    plan.keep((skip, r))
                    plan.free(skip)
                    skip = r
                y = emit_resblock(plan, P, f"upblocks.{n}", h, skip, temb, cemb, ch * mult, B, cH, cW, attn=False)
                taps[f"upblocks.{n}"] = y
                n += 1
                plan.free(h); plan.free(skip)
                h, now = y, ch * mult
            if i != 0:
                y = emit_upsample(plan, P, f"upblocks.{n}", h, B, now, cH, cW)
                taps[f"upblocks.{n}"] = y
                n += 1
                plan.free(h)
                h, cH, cW = y, 2 * cH, 2 * cW
        for skip, _, _, _ in hs:           # skips the reference never consumes
            plan.free(skip)
        sct = plan.gn_scale_shift(h, None, P["tail.0.weight"], P["tail.0.bias"], B, cH * cW)
        pk_tail = _std_pack(plan, P["tail.2.weight"], 3, 1)
        self.out_hw = (cH, cW)
        self.out = plan.buf(B, 3, cH, cW)
        plan.conv(h, None, pk_tail, P["tail.2.bias"], self.out, B=B, H=cH, W=cW, VH=cH, VW=cW, gn=sct)
        self.tail_in_src = (h, sct)        # kept for tests: the tensor entering the tail and its GN scale/shift
        plan.flush_block_vecs(vec_pos, temb, cemb, B)
        plan.keep((temb, cemb, h))
