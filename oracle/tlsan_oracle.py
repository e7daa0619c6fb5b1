"""CPU oracle for the TLSAN hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A plain numpy (float64 by default) restatement of the arithmetic of the
reference's ``TLSAN/model.py``.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product
package ``tlsan_amd`` never does (its hot path is the HIP library and it
raises when that library is missing).

PARITY PIN STATUS: **parity unpinned at the op level.**  The arithmetic of the
reference lives in TensorFlow 1.8.0 (reference README.md:6; un-vendored, no
lockfile), which is not installed in the build container and cannot be
installed (no network), and the reference ships no golden vectors or tests
(SURVEY.md section 4).  What pins this oracle instead:
  * integer / mask work: fixtures captured by importing the *real*
    ``TLSAN/input.py`` on a ``dataset.pkl`` built by the *real*
    ``TLSAN/build_dataset.py`` (tests/golden/make_fixtures.py);
  * float work: an independent second restatement (oracle/tlsan_torch_ref.py,
    torch autograd, op-for-op like the TF graph) that must agree with the
    manual backward below to 1e-10 in float64, plus finite differences;
  * end to end: README.md:34-35 AUC on the two datasets present in Data/.

Every function cites the reference lines it follows (paths relative to
/root/reference).
"""
from __future__ import annotations

import numpy as np

VERY_NEGATIVE_NUMBER = -1e30  # TLSAN/model.py:10-11

REG_TABLES = ("user_emb", "item_emb", "cate_emb", "usert_emb")  # model.py:164-169
FWA_KEYS = ("W1", "b1", "W2", "b2")


# --------------------------------------------------------------------------- params
def glorot_uniform(rng, shape):
    """TF-1.x default initializer for ``tf.get_variable`` without an initializer
    (model.py:62-64,70-72,79-81,446): glorot_uniform, limit sqrt(6/(fan_in+fan_out))."""
    fan_in, fan_out = shape[0], shape[1]
    limit = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-limit, limit, size=shape)


def init_params(config, seed=1234, dtype=np.float64):
    """Variables of model.py:58-81 + attention weights (:443-450) + tf.layers.dense (:347).

    ``gamma``=1 (:58-60), ``item_b``=0 (:65-68), ``usert_emb``=-1 (:74-77), biases 0.
    """
    rng = np.random.RandomState(seed)
    I, U, C = config["item_count"], config["user_count"], config["cate_count"]
    di, du, dc = (config["itemid_embedding_size"], config["userid_embedding_size"],
                  config["cateid_embedding_size"])
    d = config["hidden_units"]
    H = config["num_heads"]
    Ls = config["Ls"]
    assert di + dc == d and du + dc == d, "model.py:100-109,135 need d = d_i+d_c = d_u+d_c"
    assert d % H == 0
    dh = d // H
    p = {
        "gamma": np.array(1.0),
        "item_emb": glorot_uniform(rng, (I, di)),
        "item_b": np.zeros(I),
        "user_emb": glorot_uniform(rng, (U, du)),
        "usert_emb": -np.ones((U, Ls)),
        "cate_emb": glorot_uniform(rng, (C, dc)),
        "dense_K": glorot_uniform(rng, (d, d)),
        "dense_b": np.zeros(d),
    }
    for blk in ("fwa1", "fwa2"):
        p[blk + "_W1"] = glorot_uniform(rng, (dh, dh))
        p[blk + "_b1"] = np.zeros(dh)
        p[blk + "_W2"] = glorot_uniform(rng, (dh, dh))
        p[blk + "_b2"] = np.zeros(dh)
    return {k: np.asarray(v, dtype=dtype) for k, v in p.items()}


PARAM_ORDER = (
    "gamma", "item_emb", "item_b", "user_emb", "usert_emb", "cate_emb",
    "fwa1_W1", "fwa1_b1", "fwa1_W2", "fwa1_b2", "dense_K", "dense_b",
    "fwa2_W1", "fwa2_b1", "fwa2_W2", "fwa2_b2",
)


# --------------------------------------------------------------------------- batch
def as_batch(batch, is_test=False):
    """The 9-tuple of TLSAN/input.py:54 (train) / :107 (test) -> dict of numpy arrays,
    fed exactly as model.py:210-222 / :239-262 feed the placeholders (:27-53)."""
    u, i, y_or_j, hist_i, hist_i_new, hist_t, sl, sl_new, c = batch
    out = {
        "u": np.asarray(u, np.int64),
        "i": np.asarray(i, np.int64),
        "hist_i": np.asarray(hist_i, np.int64),
        "hist_i_new": np.asarray(hist_i_new, np.int64),
        "hist_t": np.asarray(hist_t, np.float32),
        "sl": np.asarray(sl, np.int64),
        "sl_new": np.asarray(sl_new, np.int64),
        "u_cate": np.asarray(c, np.int64),
    }
    if is_test:
        out["j"] = np.asarray(y_or_j, np.int64)
    else:
        out["y"] = np.asarray(y_or_j, np.float32)
    return out


# --------------------------------------------------------------------------- forward
def dropout_scale(rate, seed, B, L, d, net, which, sample0=0):
    """The keep / drop pattern of tf.nn.dropout on the input of one linear map (model.py:428-431) as
    the scale it multiplies with: 1 / keep_prob where the element is kept, 0 where it is dropped,
    independently per (sample, position, channel).  TF's own random stream cannot be reproduced;
    the pattern is a counter-based hash, the same one the HIP kernel evaluates (tlsan_common.h
    drop_scale4): element e = ((((sample * 2 + net) * 128 + position) * 2 + which) * 256 + channel,
    h = murmur3's 32-bit finaliser of (e xor seed), kept iff h < keep_prob * 2^32.
    net: 0 long block / 1 short block; which: 0 input of map1 / 1 input of map2.  -> [B, L, d]"""
    keep = np.float32(1.0 - float(np.float32(rate)))        # the kernel's fp32 keep_prob
    b = (np.arange(B, dtype=np.uint64)[:, None, None] + np.uint64(sample0))
    l = np.arange(L, dtype=np.uint64)[None, :, None]
    c = np.arange(d, dtype=np.uint64)[None, None, :]
    e = ((((b * 2 + net) * 128 + l) * 2 + which) * 256 + c) & 0xFFFFFFFF
    h = (e ^ np.uint64(seed & 0xFFFFFFFF)).astype(np.uint64)
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & 0xFFFFFFFF
    h ^= h >> 16
    thr = min(int(float(keep) * 4294967296.0), 0xFFFFFFFF)
    return np.where(h < thr, np.float32(1.0) / keep, np.float32(0.0)).astype(np.float64)


def _fwa_forward(x, length, W1, b1, W2, b2, H, k1=None, k2=None):
    """feature_wise_attention, model.py:370-394 (bn off, relu then linear; k1 / k2: dropout scales
    of the two maps' inputs, None = keep_prob 1).

    x [B,L,d]; heads = contiguous channel blocks (tf.split axis 2, :374); softmax is over
    the sequence axis independently per (sample, head, channel) (:386)."""
    B, L, d = x.shape
    dh = d // H
    xh = x.reshape(B, L, H, dh)
    k1 = None if k1 is None else k1.reshape(B, L, H, dh).astype(x.dtype)
    k2 = None if k2 is None else k2.reshape(B, L, H, dh).astype(x.dtype)
    z1 = (xh if k1 is None else xh * k1) @ W1 + b1   # :380 (bn_dense_layer -> linear -> _linear :451; dropout :428-431)
    m1 = np.maximum(z1, 0.0)               # relu :405
    m2 = (m1 if k2 is None else m1 * k2) @ W2 + b2   # :382
    mask = (np.arange(L)[None, :] < length[:, None])            # sequence_mask :376
    m2m = m2 + (1.0 - mask[:, :, None, None]) * VERY_NEGATIVE_NUMBER  # :384, 480-483
    mx = m2m.max(axis=1, keepdims=True)
    e = np.exp(m2m - mx)
    soft = e / e.sum(axis=1, keepdims=True)  # :386
    out = (soft * xh).sum(axis=1)            # :387
    cache = dict(xh=xh, z1=z1, m1=m1, soft=soft, out=out, k1=k1, k2=k2)
    return out.reshape(B, d), soft, cache


def forward(p, item_cate_list, b, H, want_cache=False, dropout=None):
    """model.py:84-137 (+ attention_net :316-366).  Returns logits [B] and intermediates.
    dropout: None (inference / dropout_rate 0) or (rate, seed) -- training with
    config['dropout'] > 0 (model.py:116-118, 428-431), pattern of dropout_scale."""
    cat = np.asarray(item_cate_list, np.int64)
    dt = p["item_emb"].dtype
    i_emb = np.concatenate([p["item_emb"][b["i"]], p["cate_emb"][cat[b["i"]]]], -1)      # :84-86
    i_b = p["item_b"][b["i"]]                                                           # :87
    u_emb = np.concatenate([p["user_emb"][b["u"]], p["cate_emb"][b["u_cate"]]], -1)     # :93-95
    ut_raw = p["usert_emb"][b["u"]]                                                     # :98
    hist_t = b["hist_t"].astype(dt)
    s = p["gamma"] * (ut_raw * hist_t)                                                  # :100-102,109
    e_long = np.concatenate([p["item_emb"][b["hist_i"]],
                             p["cate_emb"][cat[b["hist_i"]]]], -1)                      # :105-107
    h = e_long * s[:, :, None]                                                          # :107-109
    h_new = np.concatenate([p["item_emb"][b["hist_i_new"]],
                            p["cate_emb"][cat[b["hist_i_new"]]]], -1)                   # :111-113
    B_, d_ = h.shape[0], h.shape[2]
    ks = [None] * 4
    if dropout is not None and dropout[0] > 0.0:
        rate, seed = dropout
        ks = [dropout_scale(rate, seed, B_, h.shape[1], d_, 0, 0), dropout_scale(rate, seed, B_, h.shape[1], d_, 0, 1),
              dropout_scale(rate, seed, B_, h_new.shape[1] + 1, d_, 1, 0), dropout_scale(rate, seed, B_, h_new.shape[1] + 1, d_, 1, 1)]
    long_, att0, c1 = _fwa_forward(h, b["sl"], p["fwa1_W1"], p["fwa1_b1"],
                                   p["fwa1_W2"], p["fwa1_b2"], H, ks[0], ks[1])         # :334-345
    bridge = long_ @ p["dense_K"] + p["dense_b"]                                        # :347
    enc = np.concatenate([bridge[:, None, :], h_new], 1)                                # :350
    short, att1, c2 = _fwa_forward(enc, b["sl_new"] + 1, p["fwa2_W1"], p["fwa2_b1"],
                                   p["fwa2_W2"], p["fwa2_b2"], H, ks[2], ks[3])         # :353-364
    u_t = short + u_emb                                                                 # :135
    logits = (u_t * i_emb).sum(-1) + i_b                                                # :137
    res = dict(logits=logits, u_t=u_t, att0=att0, att1=att1)
    if want_cache:
        res["cache"] = dict(i_emb=i_emb, u_emb=u_emb, ut_raw=ut_raw, hist_t=hist_t, s=s,
                            e_long=e_long, h=h, h_new=h_new, long=long_, bridge=bridge,
                            enc=enc, c1=c1, c2=c2)
    return res


def all_item_scores(p, item_cate_list, u_t):
    """model.py:89-90,140: eval_logits = u_t . [item_emb || cate_emb[cat]]^T + item_b."""
    cat = np.asarray(item_cate_list, np.int64)
    all_emb = np.concatenate([p["item_emb"], p["cate_emb"][cat]], -1)
    return u_t @ all_emb.T + p["item_b"]


# --------------------------------------------------------------------------- loss
def bce_with_logits(x, y):
    """tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*y + log1p(exp(-|x|))."""
    return np.maximum(x, 0.0) - x * y + np.log1p(np.exp(-np.abs(x)))


def l2_term(p):
    """model.py:164-169: sum of tf.nn.l2_loss = sum(t**2)/2 over the four tables."""
    return sum(0.5 * float((p[k].astype(np.float64) ** 2).sum()) for k in REG_TABLES)


def loss_fn(p, item_cate_list, b, H, reg, dropout=None):
    """model.py:171-172."""
    out = forward(p, item_cate_list, b, H, dropout=dropout)
    y = b["y"].astype(out["logits"].dtype)
    return bce_with_logits(out["logits"], y).mean() + reg * l2_term(p)


# --------------------------------------------------------------------------- backward
def _fwa_backward(dout, c, W1, W2, H):
    """Manual gradient of _fwa_forward (what tf.gradients, model.py:198, differentiates)."""
    xh, z1, m1, soft, k1, k2 = c["xh"], c["z1"], c["m1"], c["soft"], c.get("k1"), c.get("k2")
    B, L, Hh, dh = xh.shape
    doh = dout.reshape(B, 1, Hh, dh)
    dsoft = doh * xh
    dx = soft * doh
    dm2 = soft * (dsoft - (soft * dsoft).sum(axis=1, keepdims=True))
    dW2 = np.einsum("blhk,blhj->kj", m1 if k2 is None else m1 * k2, dm2)
    db2 = dm2.sum(axis=(0, 1, 2))
    dm1 = dm2 @ W2.T
    if k2 is not None:
        dm1 = dm1 * k2
    dz1 = dm1 * (z1 > 0)
    dW1 = np.einsum("blhk,blhj->kj", xh if k1 is None else xh * k1, dz1)
    db1 = dz1.sum(axis=(0, 1, 2))
    dxm = dz1 @ W1.T
    dx = dx + (dxm if k1 is None else dxm * k1)
    return dx.reshape(B, L, Hh * dh), dict(W1=dW1, b1=db1, W2=dW2, b2=db2)


def backward(p, item_cate_list, b, H, reg, dlogits=None, dropout=None, l2_extra=0.0):
    """Gradients of model.py:171-172's loss w.r.t. every trainable (model.py:198).

    l2_extra: sum of squares of regularised-table rows that are NOT in `p` (tests that run the oracle on the rows a
    batch touches, extracted from tables too large for numpy: the model depends on the other rows only through this
    number -- they add reg/2 * l2_extra to the loss (:164-172) and decay by the dense L2 gradient reg * W).

    Returns ``(loss, logits, grads, sparse)`` where ``grads[k]`` is the mathematically
    summed dense gradient (sparse gather-gradients scatter-added, plus ``reg*W`` for the
    four regularised tables) and ``sparse`` keeps what the clip-norm variants need:
    ``sparse['sq_per_use']`` = sum of squares of every per-use gather-gradient row with
    NO de-duplication (how TF-1.8 forms global_norm for IndexedSlices)."""
    cat = np.asarray(item_cate_list, np.int64)
    out = forward(p, item_cate_list, b, H, want_cache=True, dropout=dropout)
    c = out["cache"]
    logits, u_t = out["logits"], out["u_t"]
    dt = logits.dtype
    B = logits.shape[0]
    di = p["item_emb"].shape[1]
    du = p["user_emb"].shape[1]
    if dlogits is None:
        y = b["y"].astype(dt)
        bce = bce_with_logits(logits, y).mean()
        loss = bce + reg * (l2_term(p) + 0.5 * float(l2_extra))
        dlogits = (1.0 / (1.0 + np.exp(-logits)) - y) / B
    else:
        bce = loss = None
    g = {k: np.zeros_like(v) for k, v in p.items()}
    sq = {k: 0.0 for k in p}

    def scat(name, idx, rows):
        np.add.at(g[name], idx.reshape(-1), rows.reshape(-1, *g[name].shape[1:]))
        sq[name] += float((rows.astype(np.float64) ** 2).sum())

    du_t = dlogits[:, None] * c["i_emb"]                      # d/d u_t of :137
    di_emb = dlogits[:, None] * u_t
    scat("item_b", b["i"], dlogits)
    scat("item_emb", b["i"], di_emb[:, :di])
    scat("cate_emb", cat[b["i"]], di_emb[:, di:])
    scat("user_emb", b["u"], du_t[:, :du])                    # u_t = short + u_emb (:135)
    scat("cate_emb", b["u_cate"], du_t[:, du:])
    denc, g2 = _fwa_backward(du_t, c["c2"], p["fwa2_W1"], p["fwa2_W2"], H)
    for k in FWA_KEYS:
        g["fwa2_" + k] = g2[k]
    dbridge = denc[:, 0]
    dh_new = denc[:, 1:]
    scat("item_emb", b["hist_i_new"], dh_new[..., :di])
    scat("cate_emb", cat[b["hist_i_new"]], dh_new[..., di:])
    g["dense_K"] = c["long"].T @ dbridge
    g["dense_b"] = dbridge.sum(0)
    dlong = dbridge @ p["dense_K"].T
    dh, g1 = _fwa_backward(dlong, c["c1"], p["fwa1_W1"], p["fwa1_W2"], H)
    for k in FWA_KEYS:
        g["fwa1_" + k] = g1[k]
    de = dh * c["s"][:, :, None]
    ds = (dh * c["e_long"]).sum(-1)                           # [B,Ls]
    g["gamma"] = np.asarray((ds * c["ut_raw"] * c["hist_t"]).sum(), dtype=dt)
    scat("usert_emb", b["u"], ds * p["gamma"] * c["hist_t"])
    scat("item_emb", b["hist_i"], de[..., :di])
    scat("cate_emb", cat[b["hist_i"]], de[..., di:])
    sparse = dict(sq_per_use=dict(sq), g_sparse={k: g[k].copy() for k in REG_TABLES})
    for k in REG_TABLES:
        g[k] = g[k] + reg * p[k]                              # grad of reg*l2_loss (:164-172)
    return loss, logits, g, sparse


def global_norm(p, g, sparse, reg, mode="tf18", l2_extra=0.0):
    """Norm used by tf.clip_by_global_norm (model.py:201).

    ``tf18``: TF-1.8 aggregates a variable's IndexedSlices (from gathers) and dense
    (from l2_loss) gradients by converting the dense part to IndexedSlices and
    *concatenating* values; global_norm then squares the raw values: no de-duplication
    and no cross term between the sparse part and reg*W.  ``dedup``: the mathematically
    summed gradient.  Dense-only variables are identical in both."""
    tot = reg * reg * float(l2_extra)   # rows outside `p` (backward's l2_extra): their gradient is reg * W in either mode
    sparse_vars = set(REG_TABLES) | {"item_b"}
    for k in p:
        if mode == "tf18" and k in sparse_vars:
            tot += sparse["sq_per_use"][k]
            if k in REG_TABLES:
                tot += float(((reg * p[k].astype(np.float64)) ** 2).sum())
        else:
            tot += float((g[k].astype(np.float64) ** 2).sum())
    return float(np.sqrt(tot))


# TF-1.8 constructor defaults of the optimizers model.py:188-193 builds with only learning_rate set
OPT_DEFAULTS = {
    "adam": dict(beta1=0.9, beta2=0.999, epsilon=1e-8),       # tf.train.AdamOptimizer
    "rmsprop": dict(decay=0.9, momentum=0.0, epsilon=1e-10),  # tf.train.RMSPropOptimizer
    "adadelta": dict(rho=0.95, epsilon=1e-8),                 # tf.train.AdadeltaOptimizer
}


def init_opt_state(p, optimizer):
    """Slot variables as TF 1.8 creates them: Adam m, v = 0; RMSProp rms = 1, momentum = 0;
    Adadelta accum, accum_update = 0.  t = number of updates applied so far."""
    one = optimizer == "rmsprop"
    return dict(t=0,
                slot1={k: (np.ones_like(v) if one else np.zeros_like(v)) for k, v in p.items()},
                slot2={k: np.zeros_like(v) for k, v in p.items()})


def apply_optimizer(p, g, lr, optimizer, state, used_item_b=None):
    """opt.apply_gradients (model.py:204) for adam | rmsprop | adadelta on CLIPPED gradients g
    (TF 1.8 core/kernels/training_ops.cc: ApplyAdam, ApplyRMSProp, ApplyAdadelta; python adam.py
    _apply_sparse_shared for IndexedSlices).  The four regularised tables receive IndexedSlices that
    cover every row (gathers + dense L2 term), so they are updated everywhere; item_b only receives
    the gathered rows: the sparse RMSProp / Adadelta kernels touch those rows only (`used_item_b`,
    boolean mask), sparse Adam decays m and v of every row and updates every row."""
    hp = OPT_DEFAULTS[optimizer]
    state["t"] += 1
    t = state["t"]
    newp = {}
    for k in p:
        w, gk, s1, s2 = p[k].astype(np.float64), g[k].astype(np.float64), state["slot1"][k], state["slot2"][k]
        mask = None
        if k == "item_b" and optimizer != "adam" and used_item_b is not None:
            mask = used_item_b
        if optimizer == "adam":
            b1, b2, eps = hp["beta1"], hp["beta2"], hp["epsilon"]
            alpha = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
            n1 = s1 * b1 + gk * (1.0 - b1)
            n2 = s2 * b2 + gk * gk * (1.0 - b2)
            nw = w - alpha * n1 / (np.sqrt(n2) + eps)
        elif optimizer == "rmsprop":
            rho, mom, eps = hp["decay"], hp["momentum"], hp["epsilon"]
            n1 = s1 * rho + gk * gk * (1.0 - rho)
            n2 = s2 * mom + lr * gk / np.sqrt(n1 + eps)
            nw = w - n2
        elif optimizer == "adadelta":
            rho, eps = hp["rho"], hp["epsilon"]
            n1 = s1 * rho + gk * gk * (1.0 - rho)
            upd = np.sqrt(s2 + eps) / np.sqrt(n1 + eps) * gk
            nw = w - upd * lr
            n2 = s2 * rho + upd * upd * (1.0 - rho)
        else:
            raise ValueError(optimizer)
        if mask is not None:
            n1, n2, nw = np.where(mask, n1, s1), np.where(mask, n2, s2), np.where(mask, nw, w)
        state["slot1"][k], state["slot2"][k] = n1.astype(s1.dtype), n2.astype(s2.dtype)
        newp[k] = nw.astype(p[k].dtype)
    return newp


def train_step(p, item_cate_list, b, H, reg, lr, clip=5.0, norm_mode="tf18", optimizer="sgd", opt_state=None,
               dropout=None, l2_extra=0.0):
    """One step of model.py:185-205: grads -> clip_by_global_norm(clip) -> optimizer.  The default
    'sgd' (:195) is W -= lr * g; adam | rmsprop | adadelta (:188-193) keep their slots in
    `opt_state` (init_opt_state).  Returns (loss, new_params, info)."""
    loss, logits, g, sparse = backward(p, item_cate_list, b, H, reg, dropout=dropout, l2_extra=l2_extra)
    norm = global_norm(p, g, sparse, reg, norm_mode, l2_extra=l2_extra)
    coef = clip / max(norm, clip)                             # clip_by_global_norm
    if optimizer == "sgd":
        newp = {k: (p[k] - lr * coef * g[k]).astype(p[k].dtype) for k in p}
    else:
        used = np.zeros(p["item_b"].shape, bool)
        used[b["i"]] = True                                   # item_b is gathered by the candidates only (:87)
        newp = apply_optimizer(p, {k: coef * g[k] for k in g}, lr, optimizer, opt_state, used)
    return loss, newp, dict(norm=norm, coef=coef, grads=g, logits=logits)


# --------------------------------------------------------------------------- eval
def eval_auc_batch(p, item_cate_list, tb, H):
    """model.py:237-263: mean(logit(i_pos) - logit(j_neg) > 0), ties count as wrong."""
    bp = dict(tb)
    res1 = forward(p, item_cate_list, bp, H)["logits"]
    bn = dict(tb)
    bn["i"] = tb["j"]
    res2 = forward(p, item_cate_list, bn, H)["logits"]
    return float(np.mean(res1 - res2 > 0)), res1, res2


def label_ranks(scores, labels):
    """Rank of the label inside its row of eval_logits under TF top_k's order (higher
    score first, ties -> lower index first): what precision_at_k / recall_at_k with a
    single label per row reduce to (model.py:144-156)."""
    s_lab = scores[np.arange(scores.shape[0]), labels]
    idx = np.arange(scores.shape[1])[None, :]
    ahead = (scores > s_lab[:, None]) | ((scores == s_lab[:, None]) & (idx < labels[:, None]))
    return ahead.sum(1)


def hits_at_k(scores, labels, ks=(1, 10, 20, 30, 40, 50)):
    r = label_ranks(scores, labels)
    return np.array([(r < k).sum() for k in ks], np.int64)


def topk_ids(scores, k):
    """TF top_k order: descending score, ties by ascending index."""
    order = np.lexsort((np.arange(scores.shape[1])[None, :].repeat(scores.shape[0], 0), -scores),
                       axis=1)
    return order[:, :k]
