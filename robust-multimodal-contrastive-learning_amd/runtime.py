"""Device runtime under the ViLTransformerSS mirror: flat parameter arenas in HBM, per-batch
activation stashes / workspaces, and thin wrappers over the C ABI (include/rmcl.h).

Data layout in HBM (DESIGN.md "Data layout"):
  * q32 / k32     : fp32 parameter arenas (query / momentum), offsets from rmcl_param_layout();
                    the momentum update, AdamW and the gradient all-reduce are single passes
                    over these flat buffers instead of 161 per-tensor ops.
  * q_lp / k_lp   : bf16 shadows of the GEMM weights (bf16 mode only).
  * g32           : fp32 gradient arena, same offsets; the backward kernels accumulate into it.
  * images are converted ONCE per step to patch rows [B*144, 3072] (the K-order of the
    patch-embedding GEMM); delta, the PGD gradient and the attacked view stay in that layout.
torch is used for allocation, streams and torch.distributed only.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Tuple

import torch

from . import _lib as L
from ._lib import lib, check, P, I64, F


def stream_ptr() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def make_dims(cfg: dict, B: int, dtype: int, exact: bool, P: int = None) -> L.Dims:
    """P: image patches per sample in this pass (default: the full grid); the position table always has the full grid."""
    ps = cfg["patch_size"]
    g = cfg["image_size"] // ps
    return L.Dims(B=B, L=cfg["max_text_len"], P=g * g if P is None else P, D=cfg["hidden_size"], H=cfg["num_heads"],
                  layers=cfg["num_layers"], mlp=cfg["hidden_size"] * cfg["mlp_ratio"], patch_k=3 * ps * ps,
                  proj=128, vocab=cfg["vocab_size"], dtype=dtype, exact=int(exact), Pp=g * g)


def param_specs(cfg: dict, lay: L.Layout) -> List[Tuple[str, int, Tuple[int, ...]]]:
    """(reference state-dict name, arena element offset, shape) for every query-side parameter."""
    D = cfg["hidden_size"]
    ps = cfg["patch_size"]
    g = cfg["image_size"] // ps
    Hm = D * cfg["mlp_ratio"]
    s = [
        ("text_embeddings.word_embeddings.weight", lay.word, (cfg["vocab_size"], D)),
        ("text_embeddings.position_embeddings.weight", lay.pos, (cfg["max_text_len"], D)),
        ("text_embeddings.token_type_embeddings.weight", lay.btype, (2, D)),
        ("text_embeddings.LayerNorm.weight", lay.eln_w, (D,)),
        ("text_embeddings.LayerNorm.bias", lay.eln_b, (D,)),
        ("token_type_embeddings.weight", lay.vtype, (2, D)),
        ("transformer.cls_token", lay.cls, (1, 1, D)),
        ("transformer.pos_embed", lay.pos_img, (1, g * g + 1, D)),
        ("transformer.patch_embed.proj.weight", lay.patch_w, (D, 3, ps, ps)),
        ("transformer.patch_embed.proj.bias", lay.patch_b, (D,)),
    ]
    for i in range(cfg["num_layers"]):
        b = lay.layer0 + i * lay.layer_stride
        n = f"transformer.blocks.{i}."
        s += [
            (n + "norm1.weight", b + lay.ln1_w, (D,)), (n + "norm1.bias", b + lay.ln1_b, (D,)),
            (n + "attn.qkv.weight", b + lay.qkv_w, (3 * D, D)), (n + "attn.qkv.bias", b + lay.qkv_b, (3 * D,)),
            (n + "attn.proj.weight", b + lay.proj_w, (D, D)), (n + "attn.proj.bias", b + lay.proj_b, (D,)),
            (n + "norm2.weight", b + lay.ln2_w, (D,)), (n + "norm2.bias", b + lay.ln2_b, (D,)),
            (n + "mlp.fc1.weight", b + lay.fc1_w, (Hm, D)), (n + "mlp.fc1.bias", b + lay.fc1_b, (Hm,)),
            (n + "mlp.fc2.weight", b + lay.fc2_w, (D, Hm)), (n + "mlp.fc2.bias", b + lay.fc2_b, (D,)),
        ]
    s += [
        ("transformer.norm.weight", lay.norm_w, (D,)), ("transformer.norm.bias", lay.norm_b, (D,)),
        ("moco_head.projector.0.weight", lay.mh0_w, (D, D)), ("moco_head.projector.0.bias", lay.mh0_b, (D,)),
        ("moco_head.projector.1.weight", lay.mh1_w, (D,)), ("moco_head.projector.1.bias", lay.mh1_b, (D,)),
        ("moco_head.projector.3.weight", lay.mh3_w, (128, D)),
        ("pooler.dense.weight", lay.pool_w, (D, D)), ("pooler.dense.bias", lay.pool_b, (D,)),
        ("itm_score.fc.weight", lay.itm_w, (2, D)), ("itm_score.fc.bias", lay.itm_b, (2,)),
    ]
    return s


EMA_GROUPS = ("text_embeddings", "token_type_embeddings", "transformer", "moco_head")


def bt_layout(cfg: dict, base: int):
    """BarlowTwinsHead (heads.py:88-107; widths [8192, 8192], 8192 at vilt_module.py:115, config key "barlowtwins_dims")
    appended to the parameter arena at element offset `base`: returns (rmcl_bt_head struct, specs, elements used)."""
    D = cfg["hidden_size"]
    H1, H2, H3 = cfg.get("barlowtwins_dims", (8192, 8192, 8192))
    off = base
    offs = {}
    n = "barlowtwins_head.projector."
    shapes = [("w1", n + "0.weight", (H1, D)), ("g1", n + "1.weight", (H1,)), ("b1", n + "1.bias", (H1,)),
              ("w2", n + "3.weight", (H2, H1)), ("g2", n + "4.weight", (H2,)), ("b2", n + "4.bias", (H2,)), ("w3", n + "6.weight", (H3, H2))]
    specs = []
    for key, name, shape in shapes:
        offs[key] = off
        specs.append((name, off, shape))
        cnt = 1
        for v in shape:
            cnt *= v
        off += (cnt + 63) // 64 * 64
    return L.BtHead(D=D, H1=H1, H2=H2, H3=H3, **offs), specs, off - base


class BtBuffers:
    """Per-pass buffers of the Barlow-Twins head: its stash, the projection z, d loss / dz and the distance rows."""

    def __init__(self, eng: "Engine", B: int):
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=eng.device)
        self.B = B
        self.stash = f32(int(lib.rmcl_bt_stash_floats(C.byref(eng.bt), B)))
        self.z = f32(B, eng.bt.H3)
        self.dz = f32(B, eng.bt.H3)
        self.dcls = f32(B, eng.bt.D)
        self.rows = f32(B, 3)
        self.loss2 = f32(2)


class PassBuffers:
    """Everything sized by the per-GPU batch B (allocated once, reused every step)."""

    def __init__(self, eng: "Engine", B: int, dtype=None, P=None, lane_of: "PassBuffers" = None, lane: int = 0):
        """lane_of / lane: these buffers are LANE `lane` of `lane_of` (samples [lane * B, (lane + 1) * B) of its batch): everything that is
        laid out per sample is a row-range VIEW of the parent's tensor, only the scratch of a pass (workspace, stashes, InfoNCE
        workspace, loss ring) is the lane's own - see Engine.lanes."""
        dev = eng.device
        self.dtype = eng.dtype if dtype is None else dtype       # arithmetic of the passes run through these buffers
        d = eng.dims(B, self.dtype, P)
        if lane_of is not None:
            self._init_lane(eng, d, lane_of, lane)
            return
        self.geom = None                # RaggedGeometry of a zero-padded batch (None: full-size images, dense patches)
        self.ragged = None              # the rmcl_ragged struct handed to the encoder passes (own pos_tok / dpos_tok scratch)
        self.B = B
        self.d = d
        N = d.L + 1 + d.P
        M = B * N
        u8 = lambda n: torch.empty(int(n), dtype=torch.uint8, device=dev)
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        self.workspace = u8(lib.rmcl_workspace_bytes(C.byref(d)))
        self.stash_full = u8(lib.rmcl_stash_bytes(C.byref(d), L.MODE_FULL))
        self.stash_data = u8(lib.rmcl_stash_bytes(C.byref(d), L.MODE_DATA))
        self.hstash_q = u8(lib.rmcl_heads_stash_bytes(C.byref(d)))
        self.hstash_k = u8(lib.rmcl_heads_stash_bytes(C.byref(d)))
        self.co_mask = torch.empty(B, N, dtype=torch.int32, device=dev)
        self.xn = f32(M, d.D)
        self.patches32 = f32(B * d.P, d.patch_k)
        tdt = torch.float32 if self.dtype == L.F32 else torch.bfloat16
        self.patchesT = torch.empty(B * d.P, d.patch_k, dtype=tdt, device=dev)
        self.patchesT_full = torch.empty(B * d.P, d.patch_k, dtype=tdt, device=dev)
        self.gpatch = torch.empty(B * d.P, d.patch_k, dtype=tdt, device=dev)
        self.delta = f32(B * d.P, d.patch_k)
        self.delta_prev = f32(B * d.P, d.patch_k)
        self.amax = torch.empty(64 * B, dtype=torch.int32, device=dev)      # include/rmcl.h rmcl_pgd_step: 64 partial maxima per sample
        self.cls = f32(B, d.D)
        self.q = f32(B, d.proj)
        self.k = f32(B, d.proj)
        self.dq = f32(B, d.proj)
        self.dcls = f32(B, d.D)
        self.rows = f32(B, 10)
        self.loss_ring = torch.zeros(32, dtype=torch.float32, device=dev)    # zeroed scalars for the kernels that ACCUMULATE a loss / norm:
        self.loss_i = 0                                                      # one fill per 32 uses instead of one per use (zero_scalar)
        self.loss_sum = self.loss_ring[0:1]
        self.nce_ws = u8(lib.rmcl_infonce_ws_bytes(B, I64(eng.num_negative)))
        self.text_ids = None
        self.text_mask = None
        self.pos_tok = self.dpos_tok = None      # per-sample position rows of a zero-padded batch and their gradient (lazy)
        self.drop = {L.MODE_INFER: (0, 0.0), L.MODE_DATA: (0, 0.0), L.MODE_FULL: (0, 0.0)}
        self.tail = {}                           # mode -> the last forward in that mode used the cls-only tail


    def _init_lane(self, eng: "Engine", d, par: "PassBuffers", lane: int):
        dev, B = eng.device, d.B
        N = d.L + 1 + d.P
        u8 = lambda n: torch.empty(int(n), dtype=torch.uint8, device=dev)
        rows = lambda t, per: t[lane * B * per:(lane + 1) * B * per]
        self.geom = self.ragged = None
        self.B, self.d, self.lane, self.parent = B, d, lane, par
        self.workspace = u8(lib.rmcl_workspace_bytes(C.byref(d)))
        self.stash_full = None                                   # lanes run the data-gradient passes only
        self.stash_data = u8(lib.rmcl_stash_bytes(C.byref(d), L.MODE_DATA))
        self.hstash_q = u8(lib.rmcl_heads_stash_bytes(C.byref(d)))
        self.hstash_k = u8(lib.rmcl_heads_stash_bytes(C.byref(d)))
        self.co_mask = rows(par.co_mask, 1)
        self.xn = rows(par.xn, N)
        self.patches32, self.patchesT, self.patchesT_full = rows(par.patches32, d.P), rows(par.patchesT, d.P), rows(par.patchesT_full, d.P)
        self.gpatch, self.delta, self.delta_prev = rows(par.gpatch, d.P), rows(par.delta, d.P), rows(par.delta_prev, d.P)
        self.amax = rows(par.amax, 64)
        self.cls, self.q, self.k, self.dq, self.dcls, self.rows = (rows(t, 1) for t in (par.cls, par.q, par.k, par.dq, par.dcls, par.rows))
        self.loss_ring = torch.zeros(32, dtype=torch.float32, device=dev)
        self.loss_i = 0
        self.loss_sum = self.loss_ring[0:1]
        self.nce_ws = u8(lib.rmcl_infonce_ws_bytes(B, I64(eng.num_negative)))
        self.text_ids = self.text_mask = None
        self.pos_tok = self.dpos_tok = None
        self.drop = {L.MODE_INFER: (0, 0.0), L.MODE_DATA: (0, 0.0), L.MODE_FULL: (0, 0.0)}
        self.tail = {}


class RaggedGeometry:
    """Patch selection of a zero-padded batch [B,3,Hmax,Wmax] (VisionTransformer.visual_embed, vision_transformer.py:559-651):
    sel [B, cap] int32 flat patch indices (valid patches row-major, then pads), counts [B], hw [B,2] = (x_h, x_w), n slots."""

    def __init__(self, sel, counts, hw, n, gh, gw, shape):
        self.sel, self.counts, self.hw, self.n, self.gh, self.gw, self.shape = sel, counts, hw, n, gh, gw, shape

    def take(self, owner: torch.Tensor) -> "RaggedGeometry":
        """geometry of the batch [owner[0], owner[1], ...] (candidate sentences of the text attack reuse their sample's image)"""
        return RaggedGeometry(self.sel.index_select(0, owner).contiguous(), self.counts.index_select(0, owner).contiguous(),
                              self.hw.index_select(0, owner).contiguous(), self.n, self.gh, self.gw,
                              (int(owner.numel()),) + tuple(self.shape[1:]))


class Engine:
    def __init__(self, cfg: dict, device, dtype: str = "bf16", exact: bool = False, pgd_dtype=None):
        if not torch.cuda.is_available():
            raise L.RmclError("rmcl_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU fallback")
        self.cfg = cfg
        self.device = torch.device(device)
        self.dtype = {"f32": L.F32, "fp32": L.F32, "bf16": L.BF16}[dtype]
        self.exact = bool(exact) or self.dtype == L.F32
        # precision of the PGD inner loop only: the reference forces fp32 there (attack/pgd_attack_vilt.py:141) while its
        # main forwards run under autocast; "f32" reproduces that split on a bf16 engine (exact fp32 matrix cores: a
        # measurement / parity mode, ~6x slower than bf16 PGD).  None: PGD runs in the engine's dtype.
        self.pgd_dtype = {None: None, "f32": L.F32, "fp32": L.F32, "bf16": None}[pgd_dtype] if self.dtype == L.BF16 else None
        self.num_negative = int(cfg.get("num_negative", 65536))
        d0 = self.dims(1)
        self.layout = L.Layout()
        lib.rmcl_param_layout(C.byref(d0), C.byref(self.layout))
        lay = self.layout
        z = lambda n, dt=torch.float32: torch.zeros(int(n), dtype=dt, device=self.device)
        # optional Barlow-Twins head (loss_names["barlowtwins"] > 0): appended behind the C layout's tensors, so the optimizer,
        # the gradient reduction and the state dict see it as part of the same arena
        self.bt, self.bt_specs, extra = None, [], 0
        if cfg.get("loss_names", {}).get("barlowtwins", 0) > 0:
            self.bt, self.bt_specs, extra = bt_layout(cfg, int(lay.total))
            H = self.bt
            self.bt_running = torch.cat([torch.zeros(H.H1), torch.ones(H.H1), torch.zeros(H.H2), torch.ones(H.H2), torch.zeros(H.H3),
                                         torch.ones(H.H3)]).to(self.device)      # [mean1, var1, mean2, var2, mean3, var3]
            self.bt_tracked = torch.zeros(3, dtype=torch.int64, device=self.device)
            self.bt_corr = torch.empty(H.H3, H.H3, dtype=torch.float32, device=self.device)
            self.bt_ws = torch.empty(int(lib.rmcl_bt_loss_ws_floats(H.H3)), dtype=torch.float32, device=self.device)
            self._bt_bufs = {}
        self.total = int(lay.total) + extra
        self.q32 = z(self.total)
        self.k32 = z(lay.ema_end)
        self.g32 = z(self.total)
        self.q_lp = z(self.total, torch.bfloat16) if self.dtype == L.BF16 else None
        self.k_lp = z(lay.ema_end, torch.bfloat16) if self.dtype == L.BF16 else None
        # LayerNorm folded into the qkv / fc1 GEMMs of the passes that keep no LayerNorm output (include/rmcl.h rmcl_fold):
        # W' = W * gamma (bf16) + the s / c vectors per arena, re-derived whenever the fp32 masters change
        self.fold = {}
        self.fold_stale = {"q": True, "k": True}
        if self.dtype == L.BF16 and os.environ.get("RMCL_NO_LN_FOLD", "0") != "1":
            nw, ns = lib.rmcl_ln_fold_elems(C.byref(d0), 0), lib.rmcl_ln_fold_elems(C.byref(d0), 1)
            for a in ("q", "k"):
                wf, sc = z(nw, torch.bfloat16), z(ns)
                self.fold[a] = (wf, sc, L.Fold(wf=wf.data_ptr(), sc=sc.data_ptr()))
        # transposed bf16 shadows of the layer weights for the data-gradient GEMMs (include/rmcl.h rmcl_weight_transpose_bf16)
        self.q_lpT = z(lay.total, torch.bfloat16) if (self.dtype == L.BF16 and os.environ.get("RMCL_NO_WT", "0") != "1") else None
        self.lpT_stale = True
        self.specs = param_specs(cfg, lay) + self.bt_specs
        self._bufs: Dict[tuple, PassBuffers] = {}
        self.lp_stale = True
        self.drop_p = float(cfg.get("drop_rate", 0.0))
        self.dropout_on = False            # set by the module per step (self.training)
        self.seed_base = int(cfg.get("seed", 0)) + 1
        self.pass_counter = 0
        self.pass_log = None               # a list here records every encoder pass (tests/test_path_gpu.py: the dropout masks of a whole step)
        self._lut = None                                                   # 256-entry normalisation table of the uint8 feed path
        self.side_stream = torch.cuda.Stream(device=self.device)
        self.dw_stream = torch.cuda.Stream(device=self.device)
        self.comm_stream = torch.cuda.Stream(device=self.device)     # gradient all-reduces (N > 1), gated on backward events
        if os.environ.get("RMCL_NO_DW_STREAM", "0") != "1":          # weight-gradient GEMMs concurrent with the dX chain
            lib.rmcl_set_side_stream(C.c_void_p(self.dw_stream.cuda_stream))
        # stash prefetch of the backward (rmcl_tune_set(12, mask)): on the communication stream, idle outside the gradient reduction
        pf = {"comm": self.comm_stream, "side": self.side_stream, "dw": self.dw_stream}[os.environ.get("RMCL_PREFETCH_STREAM", "comm")]
        lib.rmcl_set_prefetch_stream(C.c_void_p(pf.cuda_stream))

    # ---- geometry ------------------------------------------------------------------------------
    def dims(self, B: int, dtype=None, P=None) -> L.Dims:
        dt = self.dtype if dtype is None else dtype
        return make_dims(self.cfg, B, dt, getattr(self, "exact", False) or dt == L.F32, P)

    def bufs(self, B: int, tag: str = "moco", dtype=None, P=None) -> PassBuffers:
        """Per-(batch size, patches per sample, objective) buffers: each objective keeps its own FULL stash so that several
        task losses of one training_step can be backpropagated after all forwards have run."""
        key = (B, tag) if P is None else (B, tag, P)
        if key not in self._bufs:
            self._bufs[key] = PassBuffers(self, B, dtype, P)
        return self._bufs[key]

    def twin(self, pb: PassBuffers, tag: str, dtype=None, owner: torch.Tensor = None) -> PassBuffers:
        """Buffers of another pass over the SAME images (same patch geometry): the key encoder, another view, the fp32 PGD
        twin; with `owner` the batch [owner[i]] (text-attack candidates).  Shares nothing but the geometry."""
        B = pb.B if owner is None else int(owner.numel())
        geom = pb.geom if (pb.geom is None or owner is None) else pb.geom.take(owner)
        pv = self.bufs(B, tag, dtype, None if pb.geom is None else pb.geom.n)
        self._set_geometry(pv, geom)
        return pv

    def _set_geometry(self, pb: PassBuffers, geom):
        pb.geom = geom
        pb.ragged = None
        if geom is not None:
            g0 = self.cfg["image_size"] // self.cfg["patch_size"]
            if pb.pos_tok is None:
                pb.pos_tok = torch.empty(pb.B, pb.d.P + 1, pb.d.D, dtype=torch.float32, device=self.device)
                pb.dpos_tok = torch.empty(pb.B, pb.d.P + 1, pb.d.D, dtype=torch.float32, device=self.device)
            pb.ragged = L.Ragged(sel=geom.sel.data_ptr(), counts=geom.counts.data_ptr(), hw=geom.hw.data_ptr(),
                                 sel_ld=geom.sel.shape[1], gw=geom.gw, G0=g0, pos_tok=pb.pos_tok.data_ptr(),
                                 dpos_tok=pb.dpos_tok.data_ptr())

    @staticmethod
    def _rg(pb: PassBuffers):
        return C.byref(pb.ragged) if pb.ragged is not None else None

    def lanes(self, pb: PassBuffers, n: int = None):
        """The batch of `pb` as n independent half-size passes (or None where that does not apply).  Why: at B = 64 every launch of the
        encoder fills the chip by itself, so all CUs load, run their k-loops and reach their (HBM-bound) epilogues TOGETHER, and most
        launches are one tile per CU - nothing runs under a tile's prologue / epilogue.  Two chains of B / 2 on two HIP streams take
        half the CUs each and drift apart, so one chain's epilogues and attention kernels fall into the other's k-loops
        (tools/two_stream_test.py: 12-layer forward 2.93 ms as one chain, 2.58-2.70 ms as two).  The passes of the PGD loop are
        independent per sample (the loss couples them only through the constant 1 / B), so each lane runs its own K-step loop.
        RMCL_LANES=0 turns this off, RMCL_LANES=1 forces it for any even B (tests); default: even B >= 32, dense full-size images,
        bf16 passes.  Dropout (round 4): every lane's pass draws its own seed (`encoder_forward`), remembered with the lane's stash, so
        forward and backward of a lane agree; the masks of a lane are indexed by the lane's own rows - a different, equally valid draw
        than the one-chain step's (tests hand the materialised per-lane masks to the oracle)."""
        mode = os.environ.get("RMCL_LANES", "auto")
        n = int(os.environ.get("RMCL_LANE_COUNT", "2")) if n is None else n
        if mode == "0" or pb.B % n or pb.geom is not None or pb.dtype != L.BF16 or self.exact:
            return None
        if mode != "1" and pb.B < 32:
            return None
        ls = getattr(pb, "_lanes", None)
        if ls is None or len(ls) != n:
            ls = pb._lanes = [PassBuffers(self, pb.B // n, pb.dtype, None, lane_of=pb, lane=i) for i in range(n)]
        for i, ln in enumerate(ls):                                          # the text tensors are new every step
            ln.text_ids = pb.text_ids[i * ln.B:(i + 1) * ln.B]
            ln.text_mask = pb.text_mask[i * ln.B:(i + 1) * ln.B]
            ln.k = pb.k[i * ln.B:(i + 1) * ln.B]                             # (pb.k may have been re-pointed: compute_moco_contrastive)
        return ls

    def pgd_bufs(self, pb: PassBuffers) -> PassBuffers:
        """Buffers of the PGD inner loop: `pb` itself, or (pgd_dtype="f32" on a bf16 engine) an fp32 twin that shares the
        batch, the keys and the perturbation buffers with `pb`."""
        if self.pgd_dtype is None or self.pgd_dtype == pb.dtype:
            return pb
        pp = self.twin(pb, "pgd_f32", self.pgd_dtype)
        pp.text_ids, pp.text_mask, pp.patches32 = pb.text_ids, pb.text_mask, pb.patches32
        pp.delta, pp.delta_prev, pp.k = pb.delta, pb.delta_prev, pb.k
        return pp

    def view(self, arena: torch.Tensor, off: int, shape) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= s
        return arena[off:off + n].view(*shape)

    # ---- parameter maintenance -----------------------------------------------------------------
    def refresh_shadows(self):
        """bf16 shadows of the query / momentum weights (after load_state_dict or an optimizer step)."""
        if self.dtype == L.BF16:
            check(lib.rmcl_cast_f32(P(self.q32), P(self.q_lp), L.BF16, I64(self.q32.numel()), stream_ptr()), "cast")
            check(lib.rmcl_cast_f32(P(self.k32), P(self.k_lp), L.BF16, I64(self.k32.numel()), stream_ptr()), "cast")
        self.lp_stale = False
        self.fold_stale = {"q": True, "k": True}
        self.lpT_stale = True

    def weights_T(self):
        """transposed bf16 weight shadows of the query arena (None: the backward reads the weights as stored)"""
        if self.q_lpT is None:
            return None
        if self.lp_stale:
            self.refresh_shadows()
        if self.lpT_stale:
            d0 = self.dims(1)
            check(lib.rmcl_weight_transpose_bf16(C.byref(d0), P(self.q_lp), P(self.q_lpT), stream_ptr()), "weight_transpose")
            self.lpT_stale = False
        return self.q_lpT

    def fold_of(self, key: bool):
        """rmcl_fold of the arena a pass reads (None: separate LayerNorm kernels), refreshed if its masters changed."""
        a = "k" if key else "q"
        if a not in self.fold:
            return None
        wf, sc, st = self.fold[a]
        if self.fold_stale[a]:
            d0 = self.dims(1)
            check(lib.rmcl_ln_fold(C.byref(d0), P(self.k32 if key else self.q32), P(wf), P(sc), stream_ptr()), "ln_fold")
            self.fold_stale[a] = False
        return C.byref(st)

    def ema(self, m: float):
        n = self.layout.ema_end
        check(lib.rmcl_ema_f32(P(self.k32), P(self.q32), P(self.k_lp), F(m), I64(n), stream_ptr()), "ema")
        self.fold_stale["k"] = True

    def zero_grads(self):
        self.g32.zero_()

    # ---- per-step data -------------------------------------------------------------------------
    def patch_geometry(self, img: torch.Tensor, select: torch.Tensor = None):
        """None for a batch of full-size images (every patch of the image_size x image_size grid valid), else the
        RaggedGeometry of the zero-padded batch.  One device->host copy of B counts (n sizes the launches).
        config["dense_images"] = True skips the check for image_size x image_size inputs (synthetic benchmarks)."""
        B, Cc, Hh, Ww = img.shape
        ps = self.cfg["patch_size"]
        S = self.cfg["image_size"]
        if Cc != 3 or Hh % ps or Ww % ps:
            raise ValueError(f"images must be [B,3,H,W] with H, W multiples of patch_size={ps} (got {tuple(img.shape)}); "
                             "MinMaxResize of the reference rounds both sides to multiples of 32 (transforms/utils.py:5-26)")
        if Hh == S and Ww == S and select is None and self.cfg.get("dense_images", False):
            return None
        gh, gw = Hh // ps, Ww // ps
        if gh * gw > 1024:
            raise ValueError(f"at most 1024 patches per image ({gh}x{gw} given)")
        sel = torch.empty(B, gh * gw, dtype=torch.int32, device=self.device)
        counts = torch.empty(B, dtype=torch.int32, device=self.device)
        hw = torch.empty(B, 2, dtype=torch.int32, device=self.device)
        check(lib.rmcl_patch_select(P(img), B, 3, Hh, Ww, ps, P(sel), P(counts), P(hw), stream_ptr()), "patch_select")
        cnt = counts.cpu()
        if Hh == S and Ww == S and select is None and bool((cnt == gh * gw).all()):
            return None
        n = int(cnt.max())
        mil = self.cfg.get("max_image_len", -1)
        if isinstance(mil, int) and mil > 0:
            n = min(n, mil)                                                # vision_transformer.py:602-616
        if n + 1 + self.cfg["max_text_len"] > 256 and self.dtype == L.BF16:
            raise NotImplementedError(f"{n} image patches + text exceed the 256-token limit of the fused attention kernels")
        if select is not None:                                             # the caller's draw (parity tests: the reference's)
            sel = select.to(self.device, torch.int32).contiguous()
            assert sel.shape == (B, n), (tuple(sel.shape), (B, n))
            counts = torch.minimum(counts, torch.full_like(counts, n))
        else:
            over = (cnt > n).nonzero().flatten().tolist()
            for b in over:                                                 # more valid patches than max_image_len: the reference
                v = int(cnt[b])                                            # keeps a random subset (multinomial w/o replacement, :633-636)
                keep = torch.multinomial(torch.ones(v).float(), n).to(self.device)
                sel[b, :n] = sel[b, :v].index_select(0, keep)
            if over:
                counts = torch.minimum(counts, torch.full_like(counts, n))
        return RaggedGeometry(sel, counts, hw, n, gh, gw, (B, 3, Hh, Ww))

    def bind_batch(self, text_ids: torch.Tensor, text_mask: torch.Tensor, image, tag: str = "moco",
                   select: torch.Tensor = None) -> PassBuffers:
        """image: the float batch [B,3,H,W] of ``collate`` - or a ``Uint8Batch`` (``collate_uint8``): the decoded bytes, normalised,
        zero-padded and cut into patch rows by ONE kernel, the patch selection derived from the known extents (no selection
        launch, no device-to-host read of the counts)."""
        if hasattr(image, "tables") and hasattr(image, "data"):                # RawUint8Batch: decoded bytes, MinMaxResize still owed
            image = self.resize_raw(image)
        if hasattr(image, "sizes") and hasattr(image, "data"):
            return self._bind_uint8(text_ids, text_mask, image, tag, select)
        img = image.to(self.device, torch.float32).contiguous()
        B, Cc, Hh, Ww = img.shape
        ps = self.cfg["patch_size"]
        geom = self.patch_geometry(img, select)
        pb = self.bufs(B, tag, None, None if geom is None else geom.n)
        self._set_geometry(pb, geom)
        pb.text_ids = text_ids.to(self.device, torch.int64).contiguous()
        pb.text_mask = text_mask.to(self.device, torch.int64).contiguous()
        if geom is None:
            check(lib.rmcl_im2patch_f32(P(img), P(pb.patches32), B, 3, Hh, Ww, ps, 0, stream_ptr()), "im2patch")
        else:
            check(lib.rmcl_im2patch_sel(P(img), P(pb.patches32), P(geom.sel), P(geom.counts), geom.sel.shape[1], B, geom.n, 3, Hh, Ww,
                                        ps, 0, stream_ptr()), "im2patch_sel")
        return pb

    def _h2d(self, t: torch.Tensor) -> torch.Tensor:
        """host -> device copy that is asynchronous only out of PINNED memory (a DataLoader with pin_memory): an asynchronous copy out of
        pageable memory leaves it to the runtime when the source is read - blocking there costs microseconds and removes the question"""
        return t.to(self.device, non_blocking=bool(t.device.type == "cpu" and t.is_pinned()))

    def resize_raw(self, raw):
        """``RawUint8Batch`` (decoded bytes at their original sizes) -> ``Uint8Batch`` on the device: MinMaxResize with PIL's integer
        arithmetic in two kernel passes (include/rmcl.h rmcl_image_resize_u8); the tables come from the host (cached per size pair)."""
        from .vilt.datasets.base_dataset import Uint8Batch
        tgt, hb, hk, vb, vk = raw.tables()
        src = self._h2d(raw.data).contiguous()
        B, Hs, Ws, _ = src.shape
        Hd, Wd = vb.shape[1], hb.shape[1]                               # (the batch extent the tables were packed for)
        # the tables are freshly built pageable host arrays that die with this call: BLOCKING copies (an asynchronous copy out of
        # pageable memory may still be reading it after the arrays are gone); the image bytes themselves belong to the caller's batch
        dev = lambda t: t.to(self.device, non_blocking=False)
        ssz, dsz, hb_d, hk_d, vb_d, vk_d = dev(raw.sizes.contiguous()), dev(tgt.contiguous()), dev(hb), dev(hk), dev(vb), dev(vk)
        tmp = torch.empty(B, Hs, Wd, 3, dtype=torch.uint8, device=self.device)
        dst = torch.empty(B, Hd, Wd, 3, dtype=torch.uint8, device=self.device)
        check(lib.rmcl_image_resize_u8(P(src), P(ssz), B, Hs, Ws, P(dsz), Hd, Wd, P(hb_d), P(hk_d), hk.shape[2], P(vb_d), P(vk_d), vk.shape[2],
                                       P(tmp), P(dst), stream_ptr()), "image_resize_u8")
        out = Uint8Batch(dst, tgt)
        out.keep_alive = (raw.data, src, ssz, dsz, hb_d, hk_d, vb_d, vk_d, tmp)   # until the stream has consumed them
        return out

    def _bind_uint8(self, text_ids, text_mask, u8, tag, select) -> PassBuffers:
        from .vilt.datasets.base_dataset import select_from_sizes
        from .vilt.transforms import normalize_lut
        ps, S = self.cfg["patch_size"], self.cfg["image_size"]
        data = self._h2d(u8.data).contiguous()
        B, Hh, Ww, _ = data.shape
        sz = torch.as_tensor(u8.sizes)
        if sz.dim() != 2 or tuple(sz.shape) != (B, 2):
            raise ValueError(f"Uint8Batch.sizes must be [B, 2] = (height, width) per sample (got {tuple(sz.shape)} for a batch of {B})")
        if ps != 32 or Hh % ps or Ww % ps or bool((sz % ps != 0).any()):
            raise ValueError(f"uint8 batches need sides that are multiples of the 32-pixel patch (got {tuple(data.shape)}, sizes {sz.tolist()})")
        if bool((sz[:, 0] < 1).any()) or bool((sz[:, 0] > Hh).any()) or bool((sz[:, 1] < 1).any()) or bool((sz[:, 1] > Ww).any()):
            # the ingest kernel addresses img + ((b * Hmax + y) * Wmax + x) * 3 from these extents: an extent outside the padded
            # batch (a hand-built batch, swapped (w, h)) would read past the sample
            raise ValueError(f"Uint8Batch.sizes must satisfy 1 <= h <= {Hh}, 1 <= w <= {Ww} (got {sz.tolist()})")
        gh, gw = Hh // ps, Ww // ps
        if gh * gw > 1024:
            raise ValueError(f"at most 1024 patches per image ({gh}x{gw} given)")
        if self._lut is None:
            self._lut = normalize_lut().to(self.device)
        full = Hh == S and Ww == S and select is None and bool((u8.sizes == S).all())
        geom = None
        if not full:
            sel, counts, hw = select_from_sizes(u8.sizes, gh, gw, ps)
            n = int(counts.max())
            mil = self.cfg.get("max_image_len", -1)
            if isinstance(mil, int) and mil > 0:
                n = min(n, mil)                                            # vision_transformer.py:602-616
            if n + 1 + self.cfg["max_text_len"] > 256 and self.dtype == L.BF16:
                raise NotImplementedError(f"{n} image patches + text exceed the 256-token limit of the fused attention kernels")
            if select is not None:
                sel = select.to(torch.int32).contiguous()
                assert sel.shape == (B, n), (tuple(sel.shape), (B, n))
                counts = torch.minimum(counts, torch.full_like(counts, n))
            else:
                for b in (counts > n).nonzero().flatten().tolist():        # random subset like the reference (:633-636)
                    v = int(counts[b])
                    keep = torch.multinomial(torch.ones(v).float(), n)
                    sel[b, :n] = sel[b, :v].index_select(0, keep)
                counts = torch.minimum(counts, torch.full_like(counts, n))
            geom = RaggedGeometry(sel.to(self.device), counts.to(self.device), hw.to(self.device), n, gh, gw, (B, 3, Hh, Ww))
        pb = self.bufs(B, tag, None, None if geom is None else geom.n)
        self._set_geometry(pb, geom)
        pb.text_ids = text_ids.to(self.device, torch.int64).contiguous()
        pb.text_mask = text_mask.to(self.device, torch.int64).contiguous()
        sizes = self._h2d(u8.sizes)
        check(lib.rmcl_image_u8_to_patches(P(data), P(sizes), P(geom.sel) if geom else None, P(geom.counts) if geom else None,
                                           geom.sel.shape[1] if geom else 0, B, geom.n if geom else gh * gw, Hh, Ww, ps, P(self._lut),
                                           P(pb.patches32), stream_ptr()), "image_u8_to_patches")
        pb.keep_alive = (data, sizes, u8, getattr(u8, "keep_alive", None))      # until the stream has consumed them
        return pb

    def bind_text(self, like: PassBuffers, text_ids: torch.Tensor, text_mask: torch.Tensor, tag: str) -> PassBuffers:
        """Buffers of another objective/view for the same images (`like.patches32` is shared) with other text."""
        pv = self.twin(like, tag)
        pv.text_ids = text_ids.to(self.device, torch.int64).contiguous()
        pv.text_mask = text_mask.to(self.device, torch.int64).contiguous()
        pv.patches32 = like.patches32
        return pv

    def patches_to_image(self, pat: torch.Tensor, pb: PassBuffers) -> torch.Tensor:
        """patch rows (delta, gradients) of `pb`'s batch back to image layout [B,3,H,W] (zero outside the selected patches)."""
        ps = self.cfg["patch_size"]
        if pb.geom is None:
            S = self.cfg["image_size"]
            out = torch.empty(pb.B, 3, S, S, dtype=torch.float32, device=self.device)
            check(lib.rmcl_im2patch_f32(P(out), P(pat), pb.B, 3, S, S, ps, 1, stream_ptr()), "patch2im")
            return out
        g = pb.geom
        out = torch.empty(g.shape, dtype=torch.float32, device=self.device)
        check(lib.rmcl_im2patch_sel(P(out), P(pat), P(g.sel), P(g.counts), g.sel.shape[1], pb.B, g.n, 3, g.shape[2], g.shape[3], ps, 1,
                                    stream_ptr()), "patch2im_sel")
        return out

    def make_operand(self, pb: PassBuffers, d1=None, d2=None, out=None) -> torch.Tensor:
        """out = cast(patches32 + d1 + d2): the `img_init + img_delta` of pgd_attack_vilt.py:144."""
        out = pb.patchesT if out is None else out
        check(lib.rmcl_add_cast_f32(P(pb.patches32), P(d1), P(d2), P(out), pb.dtype,
                                    I64(pb.patches32.numel()), stream_ptr()), "add_cast")
        return out

    # ---- encoder passes ------------------------------------------------------------------------
    def encoder_forward(self, pb: PassBuffers, key: bool, mode: int, patchesT: torch.Tensor, cls_tail: bool = False):
        """cls_tail: the caller reads only the cls row of every sample of pb.xn (the contrastive objectives: pooler -> head); the
        last block then runs its row-wise part on B rows (include/rmcl.h RMCL_MODE_CLS_TAIL).  Remembered per (buffers, mode):
        the matching encoder_backward picks the compact form by itself.  Under dropout the compact rows draw the masks of the dense
        rows they stand for (csrc/gemm.h drop_row_mul): the tail gives the dense block's numbers there too."""
        if self.lp_stale:
            self.refresh_shadows()
        tail = bool(cls_tail) and pb.B <= 1024 and os.environ.get("RMCL_NO_CLS_TAIL", "0") != "1"
        pb.tail[mode] = tail
        # dropout (reference: live in every train-mode forward incl. the key encoder and the PGD copies, SURVEY
        # quirk 6): a fresh seed per pass, remembered per stash so the matching backward regenerates the masks
        self.pass_counter += 1
        seed = (self.seed_base * 2654435761 + self.pass_counter * 40503) & 0xFFFFFFFF
        p = self.drop_p if self.dropout_on else 0.0
        pb.drop[mode] = (seed, p)
        if self.pass_log is not None:                                  # (tests: which pass drew which masks)
            self.pass_log.append({"seed": seed, "p": p, "mode": mode, "key": bool(key), "B": pb.B, "lane": getattr(pb, "lane", None), "tail": tail})
        p32, plp = (self.k32, self.k_lp) if key else (self.q32, self.q_lp)
        stash = {L.MODE_INFER: None, L.MODE_DATA: pb.stash_data, L.MODE_FULL: pb.stash_full}[mode]
        check(lib.rmcl_encoder_forward(C.byref(pb.d), mode | (L.MODE_CLS_TAIL if tail else 0), P(p32), P(plp), P(pb.text_ids), P(pb.text_mask), P(patchesT),
                                       P(pb.co_mask), P(stash), P(pb.workspace), P(pb.xn), C.c_uint32(seed), F(p), self._rg(pb),
                                       self.fold_of(key) if (mode != L.MODE_FULL and pb.dtype == L.BF16) else None, stream_ptr()),
              "encoder_forward")

    def heads_forward(self, pb: PassBuffers, key: bool, want_q: bool = True, wgrad: bool = True):
        """wgrad=False: the matching heads_backward runs with_grads=False (or not at all): the pooler input is not stashed."""
        head = self.k32 if key else self.q32
        hst = pb.hstash_k if key else pb.hstash_q
        out_q = (pb.k if key else pb.q) if want_q else None
        if not key:
            pb.heads_wgrad = bool(wgrad)
        check(lib.rmcl_heads_forward2(C.byref(pb.d), P(self.q32), P(head), P(pb.xn), P(hst), P(pb.cls), P(out_q),
                                      0 if wgrad else L.HEADS_NO_WGRAD, stream_ptr()), "heads_forward")

    @staticmethod
    def zero_scalar(pb: PassBuffers) -> torch.Tensor:
        """A zeroed 1-element fp32 tensor for a kernel that accumulates into it (``pb.loss_sum`` afterwards): the next slot of a
        pre-zeroed ring - every tiny fill launch on the step's critical path costs ~20 us."""
        if pb.loss_i == pb.loss_ring.numel():
            pb.loss_ring.zero_()
            pb.loss_i = 0
        pb.loss_sum = pb.loss_ring[pb.loss_i:pb.loss_i + 1]
        pb.loss_i += 1
        return pb.loss_sum

    def infonce(self, pb: PassBuffers, grad_scale: float, want_dq: bool, metrics: bool = True):
        """metrics=False: the caller reads only loss / dq (PGD passes) - the bf16 engine's form then skips the queue-distance sums.
        bf16 passes run the split-bf16 matrix-core form (include/rmcl.h rmcl_infonce_split_bf16), fp32 passes the exact one."""
        self.zero_scalar(pb)
        if pb.dtype == L.BF16 and not self.exact and os.environ.get("RMCL_INFONCE_EXACT", "0") != "1":
            check(lib.rmcl_infonce_split_bf16(P(pb.q), P(pb.k), P(self.queue), pb.B, 128, I64(self.num_negative),
                                              F(self.cfg["temperature"]), F(grad_scale), P(pb.dq if want_dq else None), P(pb.rows),
                                              P(pb.loss_sum), P(pb.nce_ws), 1 if metrics else 0, stream_ptr()), "infonce")
            return
        check(lib.rmcl_infonce_f32(P(pb.q), P(pb.k), P(self.queue), pb.B, 128, I64(self.num_negative),
                                   F(self.cfg["temperature"]), F(grad_scale), P(pb.dq if want_dq else None), P(pb.rows),
                                   P(pb.loss_sum), P(pb.nce_ws), stream_ptr()), "infonce")

    def heads_backward(self, pb: PassBuffers, dq, dcls_extra, with_grads: bool):
        if with_grads and not getattr(pb, "heads_wgrad", True):
            raise L.RmclError("heads_backward(with_grads=True) after heads_forward(wgrad=False): the pooler input was not stashed")
        check(lib.rmcl_heads_backward(C.byref(pb.d), P(self.q32), P(self.q32), P(pb.hstash_q), P(dq), P(dcls_extra),
                                      P(pb.dcls), P(self.g32 if with_grads else None), P(pb.workspace), stream_ptr()),
              "heads_backward")

    def encoder_backward(self, pb: PassBuffers, mode: int, patchesT, dxn, cls_only: bool, dpatches, dtext=None):
        stash = pb.stash_data if mode == L.MODE_DATA else pb.stash_full
        seed, p = pb.drop[mode]
        co = (2 if pb.tail.get(mode) else 1) if cls_only else 0
        if pb.tail.get(mode) and not cls_only:
            raise L.RmclError("encoder_backward: the forward of these buffers kept only the cls rows (cls_tail) - a full-row gradient has nowhere to go")
        check(lib.rmcl_encoder_backward(C.byref(pb.d), mode, P(self.q32), P(self.q_lp), P(pb.text_ids), P(patchesT),
                                        P(pb.co_mask), P(stash), P(pb.workspace), P(dxn), co, P(dpatches), P(dtext),
                                        P(self.g32 if mode == L.MODE_FULL else None), C.c_uint32(seed), F(p), self._rg(pb),
                                        P(self.weights_T() if pb.dtype == L.BF16 else None), stream_ptr()), "encoder_backward")

    # ---- Barlow-Twins head (include/rmcl.h rmcl_bt_*) --------------------------------------------------------------
    def bt_bufs(self, B: int, tag: str) -> BtBuffers:
        if (B, tag) not in self._bt_bufs:
            self._bt_bufs[(B, tag)] = BtBuffers(self, B)
        return self._bt_bufs[(B, tag)]

    def bt_forward(self, bb: BtBuffers, cls: torch.Tensor, training: bool, track: bool):
        """z = barlowtwins_head(cls_feats).  training: batch statistics; track: also update the module's running estimates (the
        module's own head does, the PGD / text attack's deep copy of it does not: pgd_attack_vilt.py:189)."""
        run = self.bt_running if (track or not training) else None
        check(lib.rmcl_bt_head_forward(C.byref(self.bt), P(self.q32), P(cls), bb.B, int(training), P(run), F(0.1), P(bb.stash), P(bb.z),
                                       stream_ptr()), "bt_head_forward")
        if training and track:
            self.bt_tracked += 1
        return bb.z

    def bt_backward(self, bb: BtBuffers, dz: torch.Tensor, training: bool, with_grads: bool):
        check(lib.rmcl_bt_head_backward(C.byref(self.bt), P(self.q32), P(bb.stash), P(dz), bb.B, int(training),
                                        P(self.g32 if with_grads else None), P(bb.dcls), stream_ptr()), "bt_head_backward")
        return bb.dcls

    def bt_loss(self, bb: BtBuffers, zk: torch.Tensor, denom: float, lam: float, grad_scale: float, want_dz: bool, reduce_c=None):
        """loss2 = (on_diag, off_diag) of c = z^T zk / denom (objectives.py:478-484); dz = d(grad_scale (on + lam off))/dz.
        reduce_c: callable run on the correlation matrix between its computation and the loss (the all-reduce of :480)."""
        self.bt_loss_of(bb.z, zk, bb.B, denom, lam, grad_scale, bb.loss2, reduce_c)
        if want_dz:
            check(lib.rmcl_bt_dz(P(zk), P(self.bt_corr), bb.B, self.bt.H3, F(1.0 / denom), P(bb.dz), stream_ptr()), "bt_dz")
        return bb.loss2

    def bt_loss_of(self, z: torch.Tensor, zk: torch.Tensor, B: int, denom: float, lam: float, grad_scale: float, loss2: torch.Tensor,
                   reduce_c=None):
        """(on_diag, off_diag) of c = z^T zk / denom into `loss2` (any 2-float device view); leaves the gradient matrix in
        self.bt_corr."""
        N = self.bt.H3
        check(lib.rmcl_bt_corr(P(z), P(zk), B, N, F(1.0 / denom), P(self.bt_corr), stream_ptr()), "bt_corr")
        if reduce_c is not None:
            reduce_c(self.bt_corr)
        check(lib.rmcl_bt_loss(P(self.bt_corr), N, F(lam), F(grad_scale), P(self.bt_ws), P(loss2), stream_ptr()), "bt_loss")

    def bt_pair_metrics(self, bb: BtBuffers, zk: torch.Tensor):
        check(lib.rmcl_bt_pair_metrics(P(bb.z), P(zk), bb.B, self.bt.H3, P(bb.rows), stream_ptr()), "bt_pair_metrics")
        return bb.rows

    def pgd_step(self, pb: PassBuffers, lr: float, eps: float, first: bool = False, out: torch.Tensor = None, sum_prev: bool = False):
        """delta <- clamp(delta + lr g / max|g|, +-eps) (pgd_attack_vilt.py:162-173).  ``out``: also written in the same pass,
        cast(patches32 + delta_new) = the next forward's operand (:144), or with ``sum_prev`` cast(patches32 + delta_old + delta_new) =
        the attacked view of objectives.py:176.  ``first``: the incoming delta is delta_0 = 0 and is not read (no zero fill)."""
        per = pb.d.P * pb.d.patch_k
        if out is None and not first:
            check(lib.rmcl_pgd_step(P(pb.gpatch), pb.dtype, P(pb.delta), P(pb.amax), pb.B, I64(per), F(lr), F(eps),
                                    stream_ptr()), "pgd_step")
            return
        flags = (L.PGD_DELTA_ZERO if first else 0) | (L.PGD_SUM_PREV if sum_prev else 0)
        odt = L.F32 if (out is not None and out.dtype == torch.float32) else L.BF16
        check(lib.rmcl_pgd_step_fused(P(pb.gpatch), pb.dtype, P(pb.delta), P(pb.amax), pb.B, I64(per), F(lr), F(eps), P(pb.patches32),
                                      P(out), odt, flags, stream_ptr()), "pgd_step_fused")

    def enqueue(self, keys_all: torch.Tensor, ptr: int):
        check(lib.rmcl_enqueue_f32(P(self.queue), P(keys_all), keys_all.shape[0], 128, I64(self.num_negative), I64(ptr),
                                   stream_ptr()), "enqueue")
