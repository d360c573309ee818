"""Flat parameter layout of BASIC_MODEL on the MI355X hot path and the int32 tables the native
weight-norm / packing / gradient kernels (csrc/wdsr_prep.h) walk.

All parameters of the network live in ONE fp32 buffer, in the reference's state_dict order
(models/basic_wdsr_b.py:18-83: head, body.{i}.body.{0,2,3}, tail, skip.0; each conv as bias, weight_g,
weight_v).  One buffer means one weight-norm launch, one optimizer kernel and one all-reduce message
instead of 153 of each; `state_dict()` still exposes the 153 reference keys as views into it.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass
from functools import lru_cache

import numpy as np

from . import packing as P


@dataclass(frozen=True)
class ConvSpec:
    name: str          # state_dict prefix
    cout: int
    cin: int
    k: int


class WDSRLayout:
    def __init__(self, F: int, NB: int, R: int):
        self.F, self.NB, self.R = F, NB, R
        E, L = int(F * 6), int(F * 0.84)
        self.E, self.L = E, L
        CO = 3 * R * R
        self.CO = CO
        convs = [ConvSpec("head", F, 3, 3)]
        for i in range(NB):
            convs += [ConvSpec(f"body.{i}.body.0", E, F, 1), ConvSpec(f"body.{i}.body.2", L, E, 1),
                      ConvSpec(f"body.{i}.body.3", F, L, 3)]
        convs += [ConvSpec("tail", CO, F, 3), ConvSpec("skip.0", CO, 3, 5)]
        self.convs = convs
        entries, off = OrderedDict(), 0
        for c in convs:
            for suffix, shape in (("bias", (c.cout,)), ("weight_g", (c.cout, 1, 1, 1)),
                                  ("weight_v", (c.cout, c.cin, c.k, c.k))):
                n = int(np.prod(shape))
                entries[f"{c.name}.{suffix}"] = (off, shape)
                off += n
        self.entries = entries
        self.total = off

        bg = P.BlockGeom(F, E, L)
        eg = P.EndsGeom(F, R)
        self.block_geom, self.ends_geom = bg, eg
        oh, ob, ot = eg.head_off, bg.off, eg.tail_off
        self.src_head_off = 0
        self.src_body_off = oh["size"]
        self.src_body_stride = ob["size"]
        self.src_tail_off = self.src_body_off + NB * ob["size"]
        self.src_total = self.src_tail_off + ot["size"]

        # ---- weight-norm tables ----
        chan, bias, bconst = [], [], []

        def add_conv(prefix, w_dst, b_dst):
            v_off, vshape = entries[prefix + ".weight_v"]
            g_off, _ = entries[prefix + ".weight_g"]
            cout = vshape[0]
            K = int(np.prod(vshape[1:]))
            for o in range(cout):
                chan.append((v_off + o * K, g_off + o, K, w_dst + o * K))
            if b_dst is not None:
                b_off, _ = entries[prefix + ".bias"]
                for o in range(cout):
                    bias.append((b_off + o, -1, b_dst + o))
                    bconst.append(0.0)

        add_conv("head", self.src_head_off + oh["wh"], self.src_head_off + oh["b"])
        for i in range(NB):
            base = self.src_body_off + i * ob["size"]
            add_conv(f"body.{i}.body.0", base + ob["w1"], base + ob["b1"])
            add_conv(f"body.{i}.body.2", base + ob["w2"], base + ob["b2"])
            add_conv(f"body.{i}.body.3", base + ob["w3"], base + ob["b3"])
        add_conv("tail", self.src_tail_off + ot["wt"], None)
        add_conv("skip.0", self.src_tail_off + ot["ws"], None)
        bt, _ = entries["tail.bias"]
        bs, _ = entries["skip.0.bias"]
        self.n_plain_bias = len(bias)
        for o in range(CO):                       # fused bias: tail.bias + skip.bias + image_mean
            bias.append((bt + o, bs + o, self.src_tail_off + ot["b"] + o))
            bconst.append(float("nan"))           # filled with the model's image_mean at upload time
        self.chan_tab = np.asarray(chan, dtype=np.int32)
        self.bias_tab = np.asarray(bias, dtype=np.int32)
        self.bias_const = np.asarray(bconst, dtype=np.float32)

        # ---- constants inside src ----
        ones, zeros = [self.src_head_off + oh["one"], self.src_tail_off + ot["one"]], \
                      [self.src_head_off + oh["zero"], self.src_tail_off + ot["zero"]]
        for i in range(NB):
            base = self.src_body_off + i * ob["size"]
            ones.append(base + ob["one"])
            zeros.append(base + ob["zero"])
        self.src_ones, self.src_zeros = np.asarray(ones), np.asarray(zeros)

        # ---- packing tables (int32) ----
        bt_ = P.block_tables(F, E, L)
        et_ = P.ends_tables(F, R)
        self.idx_head = et_["head"].astype(np.int32)
        self.idx_tail = et_["tail"].astype(np.int32)
        self.idx_body = bt_["w"].astype(np.int32)
        self.idx_cinit = bt_["cinit"].astype(np.int32)

        # ---- gradient gather tables: (slab index, dst index inside the layer's src vector) ----
        gt = P.block_grad_tables(F, E, L)
        n_w1, n_w2, n_w3 = E * F, L * E, F * L * 9
        ga, gb = gt["a"], gt["b"]
        dst_a = np.concatenate([ob["w1"] + np.arange(n_w1), ob["w2"] + np.arange(n_w2),
                                ob["b1"] + np.arange(E), ob["b2"] + np.arange(L)])
        dst_b = np.concatenate([ob["w3"] + np.arange(n_w3), ob["b3"] + np.arange(F)])
        def by_slab(sidx, dst):
            """in slab order: the reduction kernel then READS each of the many partial slabs in runs (coalesced) and
            scatters only its one result per element"""
            o = np.argsort(sidx, kind="stable")
            return sidx[o].astype(np.int32), dst[o].astype(np.int32)
        self.ga = by_slab(ga, dst_a)
        self.gb = by_slab(gb, dst_b)
        self.slab_a, self.slab_b = gt["a_size"], gt["b_size"]
        ge = P.ends_grad_tables(F, R)
        n_t = CO * F * 9 + CO * 75 + CO
        dst_t = np.concatenate([ot["wt"] + np.arange(CO * F * 9), ot["ws"] + np.arange(CO * 75),
                                ot["b"] + np.arange(CO)])
        assert ge["tail"].size == n_t
        self.gt = by_slab(ge["tail"], dst_t)
        dst_h = np.concatenate([oh["wh"] + np.arange(F * 27), oh["b"] + np.arange(F)])
        self.gh = by_slab(ge["head"], dst_h)
        self.slab_tail, self.slab_head = ge["tail_size"], ge["head_size"]


    def split_at(self, nb_split: int):
        """Where the late half (body[nb_split ..], tail, skip) begins: (offset in the flat parameter, row of chan_tab,
        row of bias_tab).  The flat layout and both tables are in state_dict order (head, body.0 .., tail, skip)."""
        per_block = sum(c.cout for c in self.convs if c.name.startswith("body.0."))
        head_rows = next(c.cout for c in self.convs if c.name == "head")
        first = min(off for name, (off, _) in self.entries.items() if name.startswith(f"body.{nb_split}."))
        return first, head_rows + nb_split * per_block, head_rows + nb_split * per_block


@lru_cache(maxsize=None)
def get_layout(F: int, NB: int, R: int) -> WDSRLayout:
    return WDSRLayout(F, NB, R)
