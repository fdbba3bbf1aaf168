"""torch-CPU fp32 restatement of the reference's detection hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the checker for the HIP path, never the
thing measured or shipped.  Every function cites the reference file:line it follows.
It is functional (no nn.Module): the network is a list of plain layer records produced by
``parse_cfg`` and a ``state_dict`` using the reference's key names
(``model.{i}.conv.weight``, ``model.{i}.bn.*``, ``model.{i}[.{j}].cv1.conv.weight``,
``model.{L}.m.{k}.weight|bias``, ``model.{L}.anchors``).

Pinned against the unmodified reference by tests/golden/make_golden.py (run where
/root/reference exists) -> tests/golden/*.pt, re-checked by tests/test_oracle_golden.py.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any

import torch
import torch.nn.functional as F

from . import upstream

BN_EPS = 1e-3  # reference models/yolo.py:229 via upstream initialize_weights
BN_MOMENTUM = 0.03


# =========================================================================== topology
@dataclass
class Layer:
    i: int
    f: Any  # int or list[int], reference "from"
    kind: str  # Conv | Bottleneck | SPP | Concat | Upsample | MaxPool | ZeroPad | Detect
    n: int = 1  # repeats (Bottleneck xN is a Sequential in the reference)
    c1: int = 0
    c2: int = 0
    args: list = field(default_factory=list)


def parse_cfg(d: dict, ch: int = 3, nc: int | None = None, anchors=None):
    """Topology from a model dict; follows reference models/yolo.py:298-380 (parse_model) for the
    module kinds the yolov3*.yaml files use.  Returns (layers, save, anchors, nc)."""
    d = dict(d)
    if nc is not None:
        d["nc"] = nc  # reference models/yolo.py:207-209
    if anchors is not None:
        d["anchors"] = anchors  # reference models/yolo.py:210-212
    anchors_v, nc_v = d["anchors"], d["nc"]
    gd, gw = d["depth_multiple"], d["width_multiple"]
    na = len(anchors_v[0]) // 2 if isinstance(anchors_v, list) else anchors_v
    no = na * (nc_v + 5)
    layers, save, chans, c2 = [], [], [ch], ch
    for i, (f, n, m, args) in enumerate(d["backbone"] + d["head"]):
        args = [nc_v if a == "nc" else anchors_v if a == "anchors" else (None if a == "None" else a) for a in args]
        n = max(round(n * gd), 1) if n > 1 else n  # reference models/yolo.py:325
        m = m.replace("nn.", "")
        if m in ("Conv", "Bottleneck", "SPP"):
            c1, c2 = chans[f], args[0]
            if c2 != no:
                c2 = upstream.make_divisible(c2 * gw, 8)  # reference models/yolo.py:347-348
            lay = Layer(i, f, m, n, c1, c2, list(args[1:]))
        elif m == "Concat":
            c2 = sum(chans[x] for x in f)
            lay = Layer(i, f, "Concat", 1, 0, c2, list(args))
        elif m == "Detect":
            lay = Layer(i, f, "Detect", 1, 0, 0, [args[0], args[1], [chans[x] for x in f]])
        elif m == "Upsample":
            c2 = chans[f]
            lay = Layer(i, f, "Upsample", 1, c2, c2, list(args))
        elif m == "MaxPool2d":
            c2 = chans[f]
            lay = Layer(i, f, "MaxPool", 1, c2, c2, list(args))
        elif m == "ZeroPad2d":
            c2 = chans[f]
            lay = Layer(i, f, "ZeroPad", 1, c2, c2, list(args))
        else:
            raise NotImplementedError(f"module {m} is not used by the yolov3 yamls")
        save.extend(x % i for x in ([f] if isinstance(f, int) else f) if x != -1)  # reference models/yolo.py:375
        layers.append(lay)
        if i == 0:
            chans = []
        chans.append(c2)
    return layers, sorted(save), anchors_v, nc_v


def layer_hw(layers, h, w):
    """Output (h, w) of every layer for an (h, w) input -- spatial bookkeeping only."""
    sizes = []
    cur = (h, w)
    for lay in layers:
        if lay.kind == "Detect":
            sizes.append(None)
            continue
        if isinstance(lay.f, int):
            src = cur if lay.f == -1 else sizes[lay.f]
        else:
            src = cur if lay.f[0] == -1 else sizes[lay.f[0]]
        if lay.kind == "Conv":
            k = lay.args[0] if lay.args else 1
            st = lay.args[1] if len(lay.args) > 1 else 1
            cur = tuple((v + 2 * (k // 2) - k) // st + 1 for v in src)
        elif lay.kind == "Upsample":
            cur = tuple(v * lay.args[1] for v in src)
        elif lay.kind == "MaxPool":
            k, st, p = (list(lay.args) + [None, 0])[:3]
            st = k if st is None else st
            cur = tuple((v + 2 * p - k) // st + 1 for v in src)
        elif lay.kind == "ZeroPad":
            l, r, t, b = lay.args[0]
            cur = (src[0] + t + b, src[1] + l + r)
        else:  # Bottleneck, SPP, Concat keep the spatial size
            cur = src
        sizes.append(cur)
    return sizes


def model_strides(layers, s=256):
    """Strides of the Detect inputs.  reference models/yolo.py:219-222 dry-runs a 256x256 zero
    image through the net; the same numbers fall out of the spatial bookkeeping."""
    sizes = layer_hw(layers, s, s)
    det = layers[-1]
    return [s // sizes[x][0] for x in det.f]


# =========================================================================== layers
def _conv_block(sd, prefix, x, k, s, training, stats=None):
    """Conv = act(bn(conv(x))): reference models/common.py:57-81.  Fused form (no ``bn`` keys,
    conv has bias) follows forward_fuse :77-81.  pad = k//2 (autopad :48-54)."""
    w = sd[prefix + ".conv.weight"]
    b = sd.get(prefix + ".conv.bias")
    y = F.conv2d(x, w, b, stride=s, padding=k // 2)
    if prefix + ".bn.weight" in sd:
        rm, rv = sd[prefix + ".bn.running_mean"], sd[prefix + ".bn.running_var"]
        if training:
            rm, rv = rm.clone(), rv.clone()
            y = F.batch_norm(y, rm, rv, sd[prefix + ".bn.weight"], sd[prefix + ".bn.bias"], True, BN_MOMENTUM, BN_EPS)
            if stats is not None:
                stats[prefix + ".bn.running_mean"], stats[prefix + ".bn.running_var"] = rm, rv
        else:
            y = F.batch_norm(y, rm, rv, sd[prefix + ".bn.weight"], sd[prefix + ".bn.bias"], False, BN_MOMENTUM, BN_EPS)
    return F.silu(y)


def _bottleneck(sd, prefix, x, shortcut, c1, c2, training, stats):
    """x + cv2(cv1(x)) when shortcut and c1==c2: reference models/common.py:150-165."""
    y = _conv_block(sd, prefix + ".cv1", x, 1, 1, training, stats)
    y = _conv_block(sd, prefix + ".cv2", y, 3, 1, training, stats)
    return x + y if (shortcut and c1 == c2) else y


def _spp(sd, prefix, x, ks, training, stats):
    """cv2(cat([x, mp5, mp9, mp13])): reference models/common.py:267-290."""
    x = _conv_block(sd, prefix + ".cv1", x, 1, 1, training, stats)
    pools = [F.max_pool2d(x, k, 1, k // 2) for k in ks]
    return _conv_block(sd, prefix + ".cv2", torch.cat([x] + pools, 1), 1, 1, training, stats)


def detect_decode(raw, anchors_grid, strides):
    """Eval branch of Detect.forward: reference models/yolo.py:100-110 with grids from :112-123.
    ``raw``: list of (bs,na,ny,nx,no); ``anchors_grid``: (nl,na,2) in grid units; returns (bs,sum,no).
    Arithmetic is done in raw's dtype exactly in the reference's op order."""
    z = []
    for i, x in enumerate(raw):
        bs, na, ny, nx, no = x.shape
        t = anchors_grid.dtype
        yv, xv = torch.meshgrid(torch.arange(ny, dtype=t), torch.arange(nx, dtype=t), indexing="ij")
        grid = torch.stack((xv, yv), 2).expand(1, na, ny, nx, 2) - 0.5
        agrid = (anchors_grid[i] * strides[i]).view(1, na, 1, 1, 2).expand(1, na, ny, nx, 2)
        s = x.sigmoid()
        xy = (s[..., 0:2] * 2 + grid) * strides[i]
        wh = (s[..., 2:4] * 2) ** 2 * agrid
        z.append(torch.cat((xy, wh, s[..., 4:]), 4).view(bs, na * ny * nx, no))
    return torch.cat(z, 1)


def forward(layers, save, sd, x, strides=None, training=False, stats=None):
    """Graph walk: reference models/yolo.py:135-147 (_forward_once).  Returns the Detect output:
    training -> list of raw (bs,na,ny,nx,no); eval -> (pred(bs,N,no), raw list) (models/yolo.py:110)."""
    ys = []
    for lay in layers:
        if lay.f != -1:
            x = ys[lay.f] if isinstance(lay.f, int) else [x if j == -1 else ys[j] for j in lay.f]
        p = f"model.{lay.i}"
        if lay.kind == "Conv":
            k = lay.args[0] if lay.args else 1
            s = lay.args[1] if len(lay.args) > 1 else 1
            x = _conv_block(sd, p, x, k, s, training, stats)
        elif lay.kind == "Bottleneck":
            shortcut = lay.args[0] if lay.args else True
            c1 = lay.c1
            for j in range(lay.n):
                pj = f"{p}.{j}" if lay.n > 1 else p
                x = _bottleneck(sd, pj, x, shortcut, c1, lay.c2, training, stats)
                c1 = lay.c2
        elif lay.kind == "SPP":
            x = _spp(sd, p, x, lay.args[0] if lay.args else (5, 9, 13), training, stats)
        elif lay.kind == "Concat":
            x = torch.cat(x, 1)  # reference models/common.py:428
        elif lay.kind == "Upsample":
            x = F.interpolate(x, scale_factor=float(lay.args[1]), mode=lay.args[2])
        elif lay.kind == "MaxPool":
            k, s, pd = (list(lay.args) + [None, 0])[:3]
            x = F.max_pool2d(x, k, k if s is None else s, pd)
        elif lay.kind == "ZeroPad":
            x = F.pad(x, lay.args[0])
        elif lay.kind == "Detect":
            nc, anchors, chs = lay.args
            na = len(anchors[0]) // 2
            no = nc + 5
            raw = []
            for k_, xi in enumerate(x):
                yi = F.conv2d(xi, sd[f"{p}.m.{k_}.weight"], sd[f"{p}.m.{k_}.bias"])  # models/yolo.py:96
                bs, _, ny, nx = yi.shape
                raw.append(yi.view(bs, na, no, ny, nx).permute(0, 1, 3, 4, 2).contiguous())  # :98
            if training:
                return raw
            return detect_decode(raw, sd[f"{p}.anchors"], strides), raw
        ys.append(x if lay.i in save else None)
    return x


def forward_augment(layers, save, sd, x, strides):
    """Test-time augmentation: reference models/yolo.py:239-276 (_forward_augment: scales 1 / 0.83 / 0.67, the middle pass mirrored left-right;
    _descale_pred with inplace=True; _clip_augmented).  Returns the concatenated prediction (bs, rows, no)."""
    img_size = x.shape[-2:]
    gs = int(max(float(v) for v in strides))
    nl = len(strides)
    y = []
    for si, fi in zip([1, 0.83, 0.67], [None, 3, None]):
        xi = upstream.scale_img(x.flip(fi) if fi else x, si, gs=gs)
        yi = forward(layers, save, sd, xi, strides)[0].clone()
        yi[..., :4] /= si                                   # :255
        if fi == 2:
            yi[..., 1] = img_size[0] - yi[..., 1]
        elif fi == 3:
            yi[..., 0] = img_size[1] - yi[..., 0]           # :259
        y.append(yi)
    g = sum(4**k for k in range(nl))                         # :271-276
    y[0] = y[0][:, : y[0].shape[1] - (y[0].shape[1] // g) * 1]
    y[-1] = y[-1][:, (y[-1].shape[1] // g) * 4 ** (nl - 1) :]
    return torch.cat(y, 1)


# =========================================================================== NMS
def non_max_suppression(
    prediction,
    conf_thres=0.25,
    iou_thres=0.45,
    classes=None,
    agnostic=False,
    multi_label=False,
    labels=(),
    max_det=300,
    nm=0,
    stable_sort=True,
):
    """reference utils/general.py:630-750, per image:
      :669 obj filter (strict >) -> :689-695 optional apriori label rows -> :702 cls*=obj (input dtype)
      -> :705 xywh->xyxy (input dtype) -> :709-714 multi-label rows in nonzero order / best class
      -> :717-718 class filter -> :728 descending score sort, cap 30000 -> :731-733 class-offset
      boxes (fp32) into nms -> :734 max_det -> :743 gather.
    The wall-clock guard (:675,746-748) is intentionally absent (SURVEY 8a' item 5).
    ``stable_sort=True`` makes score ties deterministic (the reference argsort is unstable, so ties
    are undefined there; see SURVEY 8a' item 3)."""
    assert 0 <= conf_thres <= 1, f"Invalid Confidence threshold {conf_thres}, valid values are between 0.0 and 1.0"
    assert 0 <= iou_thres <= 1, f"Invalid IoU {iou_thres}, valid values are between 0.0 and 1.0"
    if isinstance(prediction, (list, tuple)):
        prediction = prediction[0]
    bs = prediction.shape[0]
    nc = prediction.shape[2] - nm - 5
    max_wh, max_nms = 7680, 30000
    multi_label = multi_label and nc > 1
    mi = 5 + nc
    cand = prediction[..., 4] > conf_thres
    out = [torch.zeros((0, 6 + nm))] * bs
    for xi in range(bs):
        x = prediction[xi][cand[xi]]
        if labels and len(labels[xi]):
            lb = labels[xi]
            v = torch.zeros((len(lb), nc + nm + 5))
            v[:, :4] = lb[:, 1:5]
            v[:, 4] = 1.0
            v[range(len(lb)), lb[:, 0].long() + 5] = 1.0
            x = torch.cat((x, v), 0)
        if not x.shape[0]:
            continue
        x = x.clone()
        x[:, 5:] *= x[:, 4:5]
        box = upstream.xywh2xyxy(x[:, :4])
        mask = x[:, mi:]
        if multi_label:
            i, j = (x[:, 5:mi] > conf_thres).nonzero(as_tuple=False).T
            x = torch.cat((box[i], x[i, 5 + j, None], j[:, None].float(), mask[i]), 1)
        else:
            conf, j = x[:, 5:mi].max(1, keepdim=True)
            x = torch.cat((box, conf, j.float(), mask), 1)[conf.view(-1) > conf_thres]
        if classes is not None:
            x = x[(x[:, 5:6] == torch.tensor(classes)).any(1)]
        if not x.shape[0]:
            continue
        x = x[x[:, 4].argsort(descending=True, stable=stable_sort)[:max_nms]]
        c = x[:, 5:6] * (0 if agnostic else max_wh)
        keep = upstream.nms(x[:, :4] + c, x[:, 4], iou_thres)[:max_det]
        out[xi] = x[keep]
    return out


# =========================================================================== loss
def build_targets(shapes, targets, anchors_grid, anchor_t=4.0):
    """reference utils/loss.py:183-244.  ``shapes``: list of (bs,na,ny,nx,no) shapes;
    ``targets``: (nt,6) [img, cls, x, y, w, h] normalised; ``anchors_grid``: (nl,na,2).
    Returns per level (b, a, gj, gi, tbox(n,4), anch(n,2), tcls) with the reference's row order
    (offset-major, then anchor-major, then target order; SURVEY 8a' item 12)."""
    na, nt = anchors_grid.shape[1], targets.shape[0]
    ai = torch.arange(na).float().view(na, 1).repeat(1, nt)
    tt = torch.cat((targets.repeat(na, 1, 1), ai[..., None]), 2)  # (na,nt,7)
    g = 0.5
    off = torch.tensor([[0, 0], [1, 0], [0, 1], [-1, 0], [0, -1]]).float() * g
    out = []
    gain = torch.ones(7)
    for i, shp in enumerate(shapes):
        anchors = anchors_grid[i]
        gain[2:6] = torch.tensor(shp)[[3, 2, 3, 2]].float()
        t = tt * gain
        if nt:
            r = t[..., 4:6] / anchors[:, None]
            j = torch.max(r, 1 / r).max(2)[0] < anchor_t
            t = t[j]
            gxy = t[:, 2:4]
            gxi = gain[[2, 3]] - gxy
            j, k = ((gxy % 1 < g) & (gxy > 1)).T
            l, m = ((gxi % 1 < g) & (gxi > 1)).T
            sel = torch.stack((torch.ones_like(j), j, k, l, m))
            t = t.repeat((5, 1, 1))[sel]
            offsets = (torch.zeros_like(gxy)[None] + off[:, None])[sel]
        else:
            t = tt[0]
            offsets = 0
        bc, gxy, gwh, a = t.chunk(4, 1)
        a, (b, c) = a.long().view(-1), bc.long().T
        gij = (gxy - offsets).long()
        gi, gj = gij.T  # views: the in-place clamps below also change gij, as in the reference (:235-240),
        gj.clamp_(0, shp[2] - 1)  # so tbox is computed from the CLAMPED cell indices
        gi.clamp_(0, shp[3] - 1)
        out.append((b, a, gj, gi, torch.cat((gxy - gij, gwh), 1), anchors[a], c))
    return out


def compute_loss(p, targets, anchors_grid, hyp, nc, balance=None, autobalance_ssi=None, sort_obj_iou=False):
    """reference utils/loss.py:131-181 with criteria from :104-129 (BCEWithLogits with pos_weight,
    label smoothing, balance [4,1,0.4] for 3 levels else first nl of [4,1,.25,.06,.02]; gr=1).
    FocalLoss (:31-63) applied when hyp['fl_gamma']>0.  `balance`: the per-level objectness weights to use (a list that is UPDATED IN PLACE
    when `autobalance_ssi` is given: utils/loss.py:171-175, each level's weight moves by 1e-4 towards 1 / its objectness loss after that
    level's term was added, then all are divided by the weight of the stride-16 level `autobalance_ssi`).
    `sort_obj_iou`: ComputeLoss.sort_obj_iou (:101, :156-158; default False).
    Returns (loss(1,), items(3,), aux) where loss=(lbox+lobj+lcls)*bs (:181)."""
    nl = len(p)
    if balance is None:
        balance = {3: [4.0, 1.0, 0.4]}.get(nl, [4.0, 1.0, 0.25, 0.06, 0.02])
    cp, cn = upstream.smooth_bce(eps=hyp.get("label_smoothing", 0.0))
    cls_pw = torch.tensor([hyp["cls_pw"]])
    obj_pw = torch.tensor([hyp["obj_pw"]])
    gamma = hyp.get("fl_gamma", 0.0)

    def bce(pred, true, pw):
        if gamma > 0:  # FocalLoss around BCE(reduction='none'), alpha=0.25: utils/loss.py:31-63
            loss = F.binary_cross_entropy_with_logits(pred, true, pos_weight=pw, reduction="none")
            pp = torch.sigmoid(pred)
            p_t = true * pp + (1 - true) * (1 - pp)
            alpha_factor = true * 0.25 + (1 - true) * (1 - 0.25)
            return (loss * alpha_factor * (1.0 - p_t) ** gamma).mean()
        return F.binary_cross_entropy_with_logits(pred, true, pos_weight=pw)

    lcls, lbox, lobj = torch.zeros(1), torch.zeros(1), torch.zeros(1)
    tg = build_targets([pi.shape for pi in p], targets, anchors_grid, hyp["anchor_t"])
    aux = []
    for i, pi in enumerate(p):
        b, a, gj, gi, tbox, anch, tcls = tg[i]
        tobj = torch.zeros(pi.shape[:4], dtype=pi.dtype)
        n = b.shape[0]
        if n:
            pxy, pwh, _, pcls = pi[b, a, gj, gi].split((2, 2, 1, nc), 1)
            pxy = pxy.sigmoid() * 2 - 0.5
            pwh = (pwh.sigmoid() * 2) ** 2 * anch
            iou = upstream.bbox_iou(torch.cat((pxy, pwh), 1), tbox, CIoU=True).squeeze(-1)
            lbox = lbox + (1.0 - iou).mean()
            iou_t = iou.detach().clamp(0).type(tobj.dtype)
            if sort_obj_iou:   # utils/loss.py:156-158: written in ascending-iou order -- of several matches of one cell the largest iou stays
                j = iou_t.argsort()
                b, a, gj, gi, iou_t = b[j], a[j], gj[j], gi[j], iou_t[j]
            tobj[b, a, gj, gi] = iou_t
            if nc > 1:
                t = torch.full_like(pcls, cn)
                t[range(n), tcls] = cp
                lcls = lcls + bce(pcls, t, cls_pw)
        obji = bce(pi[..., 4], tobj, obj_pw)
        lobj = lobj + obji * balance[i]
        if autobalance_ssi is not None:
            balance[i] = balance[i] * 0.9999 + 0.0001 / obji.detach().item()
        aux.append(tobj)
    if autobalance_ssi is not None:
        ssv = balance[autobalance_ssi]
        for i in range(len(balance)):   # the whole list, including the unused tail of the 5-entry default when nl != 3 (utils/loss.py:175)
            balance[i] = balance[i] / ssv
    lbox = lbox * hyp["box"]
    lobj = lobj * hyp["obj"]
    lcls = lcls * hyp["cls"]
    bs = p[0].shape[0]
    return (lbox + lobj + lcls) * bs, torch.cat((lbox, lobj, lcls)).detach(), aux


# =========================================================================== synthetic inputs
def seeded_state_dict(layers, nc, anchors, strides, seed=0, fused=False):
    """Random-but-well-conditioned parameters with the reference's key names: conv Kaiming-uniform
    (torch default), BN gamma~U(.5,1.5), beta~N(0,.1), running_mean~N(0,.1), running_var~U(.5,1.5);
    Detect biases per reference models/yolo.py:282-292 (_initialize_biases)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}

    def conv(prefix, c1, c2, k):
        bound = 1.0 / math.sqrt(c1 * k * k)  # kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), +)
        sd[prefix + ".conv.weight"] = (torch.rand(c2, c1, k, k, generator=g) * 2 - 1) * bound
        sd[prefix + ".bn.weight"] = torch.rand(c2, generator=g) + 0.5
        sd[prefix + ".bn.bias"] = torch.randn(c2, generator=g) * 0.1
        sd[prefix + ".bn.running_mean"] = torch.randn(c2, generator=g) * 0.1
        sd[prefix + ".bn.running_var"] = torch.rand(c2, generator=g) + 0.5
        sd[prefix + ".bn.num_batches_tracked"] = torch.tensor(0)

    for lay in layers:
        p = f"model.{lay.i}"
        if lay.kind == "Conv":
            conv(p, lay.c1, lay.c2, lay.args[0] if lay.args else 1)
        elif lay.kind == "Bottleneck":
            c1 = lay.c1
            for j in range(lay.n):
                pj = f"{p}.{j}" if lay.n > 1 else p
                c_ = int(lay.c2 * 0.5)
                conv(pj + ".cv1", c1, c_, 1)
                conv(pj + ".cv2", c_, lay.c2, 3)
                c1 = lay.c2
        elif lay.kind == "SPP":
            ks = lay.args[0] if lay.args else (5, 9, 13)
            c_ = lay.c1 // 2
            conv(p + ".cv1", lay.c1, c_, 1)
            conv(p + ".cv2", c_ * (len(ks) + 1), lay.c2, 1)
        elif lay.kind == "Detect":
            ncls, anc, chs = lay.args
            na = len(anc[0]) // 2
            no = ncls + 5
            a = torch.tensor(anc).float().view(len(anc), -1, 2)
            sd[p + ".anchors"] = a / torch.tensor(strides).float().view(-1, 1, 1)  # models/yolo.py:224
            for k_, c in enumerate(chs):
                bound = 1.0 / math.sqrt(c)
                sd[f"{p}.m.{k_}.weight"] = (torch.rand(na * no, c, 1, 1, generator=g) * 2 - 1) * bound
                b = ((torch.rand(na * no, generator=g) * 2 - 1) * bound).view(na, -1)
                b[:, 4] += math.log(8 / (640 / strides[k_]) ** 2)
                b[:, 5 : 5 + ncls] += math.log(0.6 / (ncls - 0.99999))
                sd[f"{p}.m.{k_}.bias"] = b.view(-1)
    if fused:
        sd = fuse_state_dict(sd)
    return sd


def fuse_state_dict(sd):
    """Fold every BN into its conv (fp32), as reference models/yolo.py:163-172 + upstream
    fuse_conv_and_bn do; keys become ``*.conv.weight`` / ``*.conv.bias``."""
    out = {}
    for k, v in sd.items():
        if ".bn." in k:
            continue
        if k.endswith(".conv.weight") and k.replace(".conv.weight", ".bn.weight") in sd:
            p = k[: -len(".conv.weight")]
            scale = sd[p + ".bn.weight"] / torch.sqrt(BN_EPS + sd[p + ".bn.running_var"])
            w = torch.mm(torch.diag(scale), v.view(v.shape[0], -1)).view(v.shape)
            b = sd[p + ".bn.bias"] - sd[p + ".bn.weight"] * sd[p + ".bn.running_mean"] / torch.sqrt(
                sd[p + ".bn.running_var"] + BN_EPS
            )
            out[p + ".conv.weight"], out[p + ".conv.bias"] = w, b
        else:
            out[k] = v
    return out


def synth_targets(bs, nc, seed=1):
    """Targets (nt,6) [img, cls, x, y, w, h]: n_i~Poisson(7) clipped [0,30]; xy~U(.05,.95);
    wh log-uniform(.02,.6)."""
    g = torch.Generator().manual_seed(seed)
    rows = []
    for b in range(bs):
        n = int(torch.poisson(torch.tensor(7.0), generator=g).clamp(0, 30))
        if n == 0:
            continue
        cls = torch.randint(0, nc, (n, 1), generator=g).float()
        xy = torch.rand(n, 2, generator=g) * 0.9 + 0.05
        wh = torch.exp(torch.rand(n, 2, generator=g) * (math.log(0.6) - math.log(0.02)) + math.log(0.02))
        rows.append(torch.cat((torch.full((n, 1), float(b)), cls, xy, wh), 1))
    return torch.cat(rows, 0) if rows else torch.zeros(0, 6)


def _pow_int(u, k):
    """u**k by repeated multiplication (k a power of two): exact IEEE ops only, so the synthetic tensors are
    bit-identical on every CPU (vectorised exp/log/sigmoid differ in the last bit between ISAs)."""
    while k > 1:
        u = u * u
        k //= 2
    return u


def synth_predictions(bs, n_rows=25200, nc=80, img=640, seed=2, hits=0.03, n_gt=12, dtype=torch.float32):
    """Decoded prediction tensor (bs, n_rows, 5+nc) for NMS tests/bench, decoupled from the random-weight model
    (a random-weight model's objectness is ~0.003 everywhere).  Built from torch.rand and + - * / only
    (bit-reproducible across hosts).  Background rows: obj = 0.5*u^32*(0.05+0.95*v^4) (about 20 % pass conf 0.001,
    a few percent pass 0.25), class scores 0.5*u^64; a `hits` fraction of rows are jittered copies of
    n_gt ground-truth boxes with obj in [0.3,1] and the GT class score in [0.4,1].  Boxes are xywh pixels."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *shape: torch.rand(*shape, generator=g)
    out = torch.empty(bs, n_rows, 5 + nc)
    for b in range(bs):
        gt_xy = r(n_gt, 2) * (img * 0.8) + img * 0.1
        gt_wh = (0.04 + 0.46 * _pow_int(r(n_gt, 2), 2)) * img
        gt_c = torch.randint(0, nc, (n_gt,), generator=g)
        xy = r(n_rows, 2) * img
        wh = (0.02 + 0.58 * _pow_int(r(n_rows, 2), 4)) * img
        obj = 0.5 * _pow_int(r(n_rows), 32) * (0.05 + 0.95 * _pow_int(r(n_rows), 4))
        cls = 0.5 * _pow_int(r(n_rows, nc), 64)
        idx = (r(n_rows) < hits).nonzero().view(-1)
        k = torch.randint(0, n_gt, (idx.numel(),), generator=g)
        xy[idx] = gt_xy[k] + (r(idx.numel(), 2) - 0.5) * 0.3 * gt_wh[k]
        wh[idx] = gt_wh[k] * (0.8 + 0.4 * r(idx.numel(), 2))
        obj[idx] = 0.3 + 0.7 * r(idx.numel())
        cls[idx, gt_c[k]] = 0.4 + 0.6 * r(idx.numel())
        out[b, :, 0:2], out[b, :, 2:4] = xy, wh
        out[b, :, 4] = obj
        out[b, :, 5:] = cls
    return out.to(dtype)


def synth_raw_predictions(shapes, seed=31):
    """Raw head tensors (bs,na,ny,nx,no) for the loss tests: uniform in [-3, 3] from torch.rand only."""
    g = torch.Generator().manual_seed(seed)
    return [(torch.rand(*s, generator=g) * 6.0 - 3.0) for s in shapes]


# =========================================================================== output edge (val.py / detect.py after NMS)
def scale_boxes(img1_shape, boxes, img0_shape, ratio_pad=None):
    """Restates reference utils/general.py:613-626 (+ upstream clip_boxes): boxes (.., 4+) xyxy in letterboxed
    img1 space -> native img0 space, in place."""
    if ratio_pad is None:
        gain = min(img1_shape[0] / img0_shape[0], img1_shape[1] / img0_shape[1])
        pad = (img1_shape[1] - img0_shape[1] * gain) / 2, (img1_shape[0] - img0_shape[0] * gain) / 2
    else:
        gain = ratio_pad[0][0]
        pad = ratio_pad[1]
    boxes[..., [0, 2]] -= pad[0]
    boxes[..., [1, 3]] -= pad[1]
    boxes[..., :4] /= gain
    upstream.clip_boxes(boxes, img0_shape)
    return boxes


def process_batch(detections, labels, iouv):
    """Restates reference val.py:147-188: (N,6) [xyxy,conf,cls] x (M,5) [cls,xyxy] x (T,) -> bool (N,T).
    Same numpy calls as the reference (argsort()[::-1], np.unique(return_index)); exact IoU ties are therefore
    as undefined here as there beyond 16 candidate pairs."""
    import numpy as np

    correct = np.zeros((detections.shape[0], iouv.shape[0])).astype(bool)
    iou = upstream.box_iou(labels[:, 1:], detections[:, :4])
    correct_class = labels[:, 0:1] == detections[:, 5]
    for i in range(len(iouv)):
        x = torch.where((iou >= iouv[i]) & correct_class)
        if x[0].shape[0]:
            matches = torch.cat((torch.stack(x, 1), iou[x[0], x[1]][:, None]), 1).cpu().numpy()
            if x[0].shape[0] > 1:
                matches = matches[matches[:, 2].argsort()[::-1]]
                matches = matches[np.unique(matches[:, 1], return_index=True)[1]]
                matches = matches[np.unique(matches[:, 0], return_index=True)[1]]
            correct[matches[:, 1].astype(int), i] = True
    return torch.tensor(correct, dtype=torch.bool)


def synth_val_case(seed, n_lab=12, n_det=60, nc=5, img=(480, 640), dup=0.5):
    """One image's (detections (n_det,6), labels (n_lab,5)) in native pixel space for the matching tests, from torch.rand
    and + - * / only.  Labels: random boxes; detections: jittered copies of labels (IoU spread over 0.3-0.98, so every
    threshold of linspace(.5,.95,10) separates some), a `dup` share of labels is hit by several detections (the reference
    keeps the lowest-index one), some with a wrong class, plus background boxes; sorted by descending confidence as NMS
    returns them."""
    g = torch.Generator().manual_seed(seed)
    r = lambda *shape: torch.rand(*shape, generator=g)
    h, w = img
    wh = (0.08 + 0.4 * r(n_lab, 2)) * torch.tensor([w, h], dtype=torch.float32)
    c = (0.1 + 0.8 * r(n_lab, 2)) * torch.tensor([w, h], dtype=torch.float32)
    lab_xyxy = torch.cat((c - wh / 2, c + wh / 2), 1)
    lab_cls = torch.randint(0, nc, (n_lab, 1), generator=g).float()
    labels = torch.cat((lab_cls, lab_xyxy), 1)
    n_hit = int(n_det * 0.7) if n_lab else 0
    if n_lab:
        k = torch.randint(0, min(n_lab, max(1, int(n_lab * dup) + 1)), (n_hit,), generator=g) if dup > 0 else torch.randint(0, n_lab, (n_hit,), generator=g)
        k2 = torch.randint(0, n_lab, (n_hit,), generator=g)
        k = torch.where(r(n_hit) < 0.5, k, k2)
        jit = (r(n_hit, 4) - 0.5) * (0.02 + 0.5 * r(n_hit, 1)) * torch.cat((wh[k], wh[k]), 1)
        hit_box = lab_xyxy[k] + jit
        hit_cls = torch.where(r(n_hit, 1) < 0.85, lab_cls[k], torch.randint(0, nc, (n_hit, 1), generator=g).float())
    else:
        hit_box, hit_cls = torch.zeros(0, 4), torch.zeros(0, 1)
    n_bg = n_det - n_hit
    bwh = (0.05 + 0.3 * r(n_bg, 2)) * torch.tensor([w, h], dtype=torch.float32)
    bc = r(n_bg, 2) * torch.tensor([w, h], dtype=torch.float32)
    bg_box = torch.cat((bc - bwh / 2, bc + bwh / 2), 1)
    bg_cls = torch.randint(0, nc, (n_bg, 1), generator=g).float()
    box = torch.cat((hit_box, bg_box), 0)
    cls = torch.cat((hit_cls, bg_cls), 0)
    conf = 0.05 + 0.95 * r(n_det, 1)
    det = torch.cat((box, conf, cls), 1)
    det = det[det[:, 4].argsort(descending=True, stable=True)]
    return det.contiguous(), labels.contiguous()


def synth_ap_stats(seed, n_det=600, n_lab=200, nc=6, n_iou=10, absent=1):
    """Seeded (tp, conf, pred_cls, target_cls) NumPy statistics as val.py:416 hands them to ap_per_class: tp is monotone over the IoU
    thresholds (a detection correct at 0.75 is correct at 0.5), the last `absent` classes have labels but no predictions and class 0
    has predictions only -- the edge cases of the per-class loop (utils/metrics.py:53-58)."""
    g = torch.Generator().manual_seed(seed)
    conf = torch.rand(n_det, generator=g)
    q = torch.rand(n_det, generator=g) * (0.3 + 0.7 * conf)          # better-scored detections hit more often
    thr = torch.linspace(0.25, 0.9, n_iou)
    tp = q[:, None] > thr[None, :]
    pred_cls = torch.randint(0, max(1, nc - absent), (n_det,), generator=g).float()
    target_cls = torch.randint(1, nc, (n_lab,), generator=g).float()
    return tp.numpy(), conf.numpy(), pred_cls.numpy(), target_cls.numpy()


def synth_scale_case(img1_shape, seed=12, n=400):
    """(n, 6) rows [x1,y1,x2,y2,conf,cls] in letterboxed img1 space (boxes partly outside the image so that the clip
    matters), from synth_predictions: the scale_boxes test input."""
    pred = synth_predictions(bs=1, n_rows=n, nc=3, img=img1_shape[1], seed=seed)[0]
    return torch.cat((pred[:, 0:2] - pred[:, 2:4] / 2, pred[:, 0:2] + pred[:, 2:4] / 2, pred[:, 4:6]), 1).contiguous()


# =========================================================================== input edge (letterbox)
def resize_linear_u8(im, new_w, new_h):
    """cv2.resize(im, (new_w, new_h), interpolation=cv2.INTER_LINEAR) for uint8 HWC images, restated from OpenCV's
    resize.cpp (8-bit fixed-point path: HResizeLinear + VResizeLinear<uchar,int,short>).  cv2 itself is absent in this
    image (un-vendored dependency, `opencv-python>=4.1.1` in the reference's requirements.txt): PARITY UNPINNED.
    Anchor: reference utils/augmentations.py:130."""
    import numpy as np

    h0, w0 = im.shape[:2]
    sx_scale, sy_scale = 1.0 / (new_w / w0), 1.0 / (new_h / h0)

    def coefs(n, scale, size, clamp_frac):
        f = ((np.arange(n, dtype=np.float64) + 0.5) * scale - 0.5).astype(np.float32)
        s = np.floor(f).astype(np.int64)
        f = f - s.astype(np.float32)
        if clamp_frac:
            lo, hi = s < 0, s >= size - 1
            f = np.where(lo | hi, np.float32(0), f)
            s = np.where(lo, 0, np.where(hi, size - 1, s))
        c0 = np.rint((np.float32(1) - f) * np.float32(2048)).astype(np.int64)
        c1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return s, c0, c1

    sx, a0, a1 = coefs(new_w, sx_scale, w0, True)
    sy, b0, b1 = coefs(new_h, sy_scale, h0, False)
    sx1 = np.minimum(sx + 1, w0 - 1)
    r0, r1 = np.clip(sy, 0, h0 - 1), np.clip(sy + 1, 0, h0 - 1)
    src = im.astype(np.int64)
    hor = src[:, sx] * a0[None, :, None] + src[:, sx1] * a1[None, :, None]          # (h0, new_w, c) int
    out = (((b0[:, None, None] * (hor[r0] >> 4)) >> 16) + ((b1[:, None, None] * (hor[r1] >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def letterbox(im, new_shape=(640, 640), color=(114, 114, 114), auto=False, scaleup=True, stride=32):
    """Restates reference utils/augmentations.py:104-134 (cv2.resize / cv2.copyMakeBorder restated, see above)."""
    import numpy as np

    shape = im.shape[:2]
    if isinstance(new_shape, int):
        new_shape = (new_shape, new_shape)
    r = min(new_shape[0] / shape[0], new_shape[1] / shape[1])
    if not scaleup:
        r = min(r, 1.0)
    ratio = r, r
    new_unpad = round(shape[1] * r), round(shape[0] * r)
    dw, dh = new_shape[1] - new_unpad[0], new_shape[0] - new_unpad[1]
    if auto:
        dw, dh = np.mod(dw, stride), np.mod(dh, stride)
    dw /= 2
    dh /= 2
    if shape[::-1] != new_unpad:
        im = resize_linear_u8(im, new_unpad[0], new_unpad[1])
    top, bottom = round(dh - 0.1), round(dh + 0.1)
    left, right = round(dw - 0.1), round(dw + 0.1)
    out = np.empty((im.shape[0] + top + bottom, im.shape[1] + left + right, im.shape[2]), dtype=np.uint8)
    out[...] = np.asarray(color, dtype=np.uint8)
    out[top : top + im.shape[0], left : left + im.shape[1]] = im
    return out, ratio, (dw, dh)


def synth_image_u8(h, w, seed=0):
    """Smooth-ish uint8 RGB test image (h, w, 3): sums of coarse random blocks at three scales + fine noise, integer math only."""
    import numpy as np

    g = torch.Generator().manual_seed(seed)
    acc = torch.zeros(h, w, 3)
    for cell, amp in ((64, 110), (16, 80), (4, 40), (1, 25)):
        gh, gw = (h + cell - 1) // cell, (w + cell - 1) // cell
        blk = torch.randint(0, amp + 1, (gh, gw, 3), generator=g).float()
        acc += blk.repeat_interleave(cell, 0).repeat_interleave(cell, 1)[:h, :w]
    return acc.clamp(0, 255).to(torch.uint8).numpy()
