"""ReferenceAttentionControl - the ReferenceNet write / Backbone read bank protocol
(magicanimate/models/mutual_self_attention.py:128-159,186-317,532-543,577-627).

The reference monkey-patches BasicTransformerBlock.forward; here the UNet consults the control object
attached to it.  Same constructor arguments, `.update(writer, dtype)`, `.clear()`, and per-block banks
paired in the reference's order (torch_dfs order stable-sorted by descending width).

Read path on MI355X: the bank K/V projections are computed ONCE per (block, bank row) and the attention
kernel takes them as a second KV segment shared by the F frames of a clip - the reference's
`bank.unsqueeze(1).repeat(1, F, 1, 1)` + `torch.cat([x, bank])` (:238-241) are never materialised.
"""
from __future__ import annotations

import torch

from . import ops


class ReferenceAttentionControl:
    def __init__(self, unet, mode="write", do_classifier_free_guidance=False,
                 attention_auto_machine_weight=float("inf"), gn_auto_machine_weight=1.0, style_fidelity=1.0,
                 reference_attn=True, reference_adain=False, fusion_blocks="midup", batch_size=1):
        assert mode in ["read", "write"]
        assert fusion_blocks in ["midup", "full"]
        if reference_adain:
            # mutual_self_attention.py:319-530,544-575: the AdaIN forwards are bound to diffusers' 2-D block classes (isinstance checks,
            # :565-572) - on the 3-D Backbone only the mid block gets one, whose var_mean over dims (2, 3) of a 5-D tensor and 4-D banks from
            # the 2-D writer do not broadcast: the switch does not run upstream on this pipeline's reader either
            raise NotImplementedError("reference_adain (GroupNorm AdaIN hacks, mutual_self_attention.py:319-530) is off by default, bound to "
                                      "diffusers' 2-D block classes upstream, and outside the hot path")
        self.unet = unet
        self.mode = mode
        self.do_classifier_free_guidance = do_classifier_free_guidance
        self.reference_attn = reference_attn
        self.reference_adain = reference_adain
        self.fusion_blocks = fusion_blocks
        self.batch_size = batch_size
        self.order = unet.bank_order(fusion_blocks) if reference_attn else []
        self.bank = {p: [] for p in self.order}   # per block: list of (B_ref, L, C) tensors
        self.kv_cache, self.kv_row, self.uc_units = None, None, 0
        unet._reference_control = self

    def set_projected_banks(self, cache, row_index, uc_units):
        """Fast path of the sampling loop (read mode).  `cache`: {block prefix: (K rows, V^T, L)} - the bank K / V^T
        projections of a whole GROUP of timesteps, resident in HBM; `row_index`: device int32 word naming the row (timestep)
        in use; `uc_units`: leading batch rows of the next UNet call that skip the bank (uncond units, :243-256).  The
        per-step `update()` / bank lists are bypassed; pass cache=None to return to them."""
        self.kv_cache, self.kv_row, self.uc_units = cache, row_index, uc_units

    # ---- hooks called by UNet3DConditionModel.forward
    def _prepare(self, c, unet):
        if not self.reference_attn:
            return
        c.active = set(self.order)
        c.bank_mode = self.mode
        if self.mode == "read" and self.kv_cache is not None:
            c.bank_kv, c.bank_row = self.kv_cache, self.kv_row
            c.uc_batches = self.uc_units * c.F
            return
        if self.mode == "read":
            nb = c.B * c.F
            c.uc_batches = (c.B // 2) * c.F if self.do_classifier_free_guidance else 0  # uc_mask (:186-197,245-250)
            c.banks, c.bank_rows = {}, 0
            for p in self.order:
                lst = self.bank[p]
                if not lst:
                    continue
                t = lst[0] if len(lst) == 1 else torch.cat(lst, dim=1)  # cat of several writes along tokens (:239)
                # Under CFG the bank row of an uncond batch is dead: hidden_states_c of the uc rows is overwritten by the
                # bank-free uc path (:243-256).  A writer that ran on the cond images only hands over B/2 rows.
                c.bank_skip = 0
                if t.shape[0] < c.B:
                    if self.do_classifier_free_guidance and t.shape[0] == c.B - c.B // 2:
                        c.bank_skip = c.B // 2
                    else:
                        raise ValueError(f"bank has {t.shape[0]} rows but the UNet batch is {c.B}")
                c.bank_rows = t.shape[0]
                c.banks[p] = t.reshape(-1, t.shape[-1])
            _ = nb

    def _finish(self, c, unet):
        if self.mode == "write" and self.reference_attn:
            for p in self.order:
                if p in c.written:
                    rows = c.written[p]
                    n_ref = c.B * c.F
                    self.bank[p].append(rows.reshape(n_ref, -1, rows.shape[-1]))  # bank.append(norm_hidden_states) (:230)
        elif self.mode == "read" and self.reference_attn:
            for p in self.order:  # the hacked forward does self.bank.clear() after use (:258)
                self.bank[p] = []

    # ---- public protocol
    def update(self, writer, dtype=torch.float16):
        """reader.bank = [v.clone().to(dtype) for v in writer.bank] (:577-589): banks are rounded through
        `dtype` (fp16 by default, even in an fp32 run) and handed over in pairing order."""
        if not self.reference_attn:
            return
        if dtype not in (torch.float16, torch.float32, torch.bfloat16):
            raise ValueError(dtype)
        tgt = self.unet.dtype
        for pr, pw in zip(self.order, writer.order):
            out = []
            for v in writer.bank[pw]:
                if dtype == torch.float16:
                    out.append(ops.convert(v, tgt, fp16_round=True))
                elif dtype == torch.bfloat16:
                    out.append(ops.convert(ops.convert(v, torch.bfloat16), tgt))
                else:
                    out.append(ops.convert(v, tgt))
            self.bank[pr] = out

    def clear(self):
        for p in self.order:
            self.bank[p] = []
