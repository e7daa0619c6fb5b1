"""Second, independent CPU restatement of TLSAN/model.py -- TEST INFRASTRUCTURE ONLY.

Op-for-op like the TF-1.8 graph (separate gathers / concats / tile / head split+concat /
flattened [H*B*L, dh] matmuls / additive -1e30 mask / softmax / reduce_sum), differentiated
by torch autograd, with the reference's *dense* L2 over whole tables, global-norm clip and
plain SGD.  Two uses, both allowed by the oracle rule (tests/ and bench.py's cpu_baseline):
  1. cross-check of oracle/tlsan_oracle.py's hand-written backward (float64, 1e-10);
  2. the "port" CPU baseline timed by bench.py (float32, eager, host cores).
PARITY PIN STATUS: parity unpinned (see oracle/tlsan_oracle.py header) -- TF 1.8 is not
runnable here, so this file restates model.py from reading it.  Never imported by tlsan_amd.
"""
from __future__ import annotations

import numpy as np
import torch

VERY_NEGATIVE_NUMBER = -1e30  # model.py:10-11


def params_to_torch(p, dtype=torch.float64, requires_grad=True):
    return {k: torch.tensor(np.asarray(v), dtype=dtype, requires_grad=requires_grad)
            for k, v in p.items()}


def batch_to_torch(b, dtype=torch.float64):
    out = {}
    for k, v in b.items():
        v = np.asarray(v)
        if v.dtype.kind == "f":
            out[k] = torch.tensor(v, dtype=dtype)
        else:
            out[k] = torch.tensor(v, dtype=torch.int64)
    return out


def feature_wise_attention(rep_tensor, rep_length, num_heads, W1, b1, W2, b2, k1=None, k2=None):
    """model.py:370-394 with helpers :397-483, written the way the graph is built.  k1 / k2
    ([B, L, d] scales 0 | 1/keep_prob): what tf.nn.dropout multiplies the two maps' inputs with (:428-431)."""
    fold = lambda k: torch.cat(torch.split(k, k.shape[2] // num_heads, dim=2), dim=0)
    # :374  heads folded into batch, head-major
    rep = torch.cat(torch.split(rep_tensor, rep_tensor.shape[2] // num_heads, dim=2), dim=0)
    sl = rep.shape[1]
    rep_mask = torch.arange(sl)[None, :] < rep_length[:, None]          # :376
    rep_mask = rep_mask.repeat(num_heads, 1)                            # :377
    ivec = rep.shape[2]
    flat = (rep if k1 is None else rep * fold(k1)).reshape(-1, ivec)    # flatten :457-463 (dropout :428-431)
    map1 = torch.relu(flat @ W1 + b1).reshape(rep.shape)                # :380, 451, 466-477
    map1d = map1 if k2 is None else map1 * fold(k2)
    map2 = (map1d.reshape(-1, ivec) @ W2 + b2).reshape(rep.shape)       # :382
    map2_masked = map2 + (1 - rep_mask[:, :, None].to(map2.dtype)) * VERY_NEGATIVE_NUMBER  # :480-483
    soft = torch.softmax(map2_masked, dim=1)                            # :386
    attn = (soft * rep).sum(dim=1)                                      # :387
    attn = torch.cat(torch.split(attn, attn.shape[0] // num_heads, dim=0), dim=1)  # :388
    return attn, soft


def forward(p, item_cate_list, b, H, ks=(None, None, None, None)):
    """model.py:84-137.  ks: dropout scales of (long map1, long map2, short map1, short map2) inputs."""
    cat = torch.as_tensor(np.asarray(item_cate_list), dtype=torch.int64)
    i_emb = torch.cat([p["item_emb"][b["i"]], p["cate_emb"][cat[b["i"]]]], -1)
    i_b = p["item_b"][b["i"]]
    u_emb = torch.cat([p["user_emb"][b["u"]], p["cate_emb"][b["u_cate"]]], -1)
    ut_emb = p["usert_emb"][b["u"]]
    d = i_emb.shape[-1]
    ut_emb = (ut_emb * b["hist_t"]).unsqueeze(-1).repeat(1, 1, d)       # tile :100-102
    h_emb = torch.cat([p["item_emb"][b["hist_i"]], p["cate_emb"][cat[b["hist_i"]]]], -1)
    h_emb = h_emb * (p["gamma"] * ut_emb)                               # :107-109
    h_new = torch.cat([p["item_emb"][b["hist_i_new"]], p["cate_emb"][cat[b["hist_i_new"]]]], -1)
    enc, att0 = feature_wise_attention(h_emb, b["sl"], H, p["fwa1_W1"], p["fwa1_b1"],
                                       p["fwa1_W2"], p["fwa1_b2"], ks[0], ks[1])
    enc = (enc @ p["dense_K"] + p["dense_b"]).unsqueeze(1)              # :347
    enc = torch.cat([enc, h_new], 1)                                    # :350
    enc_new, att1 = feature_wise_attention(enc, b["sl_new"] + 1, H, p["fwa2_W1"], p["fwa2_b1"],
                                           p["fwa2_W2"], p["fwa2_b2"], ks[2], ks[3])
    u_t = enc_new + u_emb                                               # :135
    logits = (u_t * i_emb).sum(-1) + i_b                                # :137
    return logits, u_t


def loss_fn(p, item_cate_list, b, H, reg, ks=(None, None, None, None)):
    """model.py:164-172."""
    logits, _ = forward(p, item_cate_list, b, H, ks)
    l2 = sum(0.5 * (p[k] ** 2).sum() for k in ("user_emb", "item_emb", "cate_emb", "usert_emb"))
    bce = torch.nn.functional.binary_cross_entropy_with_logits(logits, b["y"].to(logits.dtype))
    return bce + reg * l2, logits


def grads(p, item_cate_list, b, H, reg, ks=(None, None, None, None)):
    names = list(p.keys())
    loss, logits = loss_fn(p, item_cate_list, b, H, reg, ks)
    gs = torch.autograd.grad(loss, [p[k] for k in names])
    return loss.detach(), logits.detach(), dict(zip(names, gs))


def train_step_(p, item_cate_list, b, H, reg, lr, clip=5.0):
    """In-place SGD step (model.py:185-205, 'sgd'), clip norm over the *summed* gradients
    (the 'dedup' variant; autograd has already summed duplicate gathers)."""
    loss, logits, g = grads(p, item_cate_list, b, H, reg)
    norm = torch.sqrt(sum((v.double() ** 2).sum() for v in g.values()))
    coef = clip / max(float(norm), clip)
    with torch.no_grad():
        for k in p:
            p[k] -= lr * coef * g[k]
    return float(loss), float(norm)
