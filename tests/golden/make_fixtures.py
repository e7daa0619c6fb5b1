#!/usr/bin/env python3
"""Generate the committed golden fixtures.  RUNS ONLY IN THE BUILD CONTAINER, where
/root/reference exists; nothing here is imported by tests or by the product.

What it does (SURVEY.md section 8c "What pins results"):
  1. executes the reference's *real* ``TLSAN/build_dataset.py`` (in a scratch dir under /tmp,
     with ``../Data`` symlinked to /root/reference/Data; for Clothing the dataset path string is
     swapped in memory, as the reference's README asks users to do by hand) to obtain
     ``dataset.pkl`` = train_set, test_set, (U,I,C), item_cate_list;
  2. imports the *real* ``TLSAN/input.py`` and records what ``DataInput`` / ``DataInputTest``
     emit for chosen batches -> ``batches_<name>.npz`` (bit-exact integer/mask pins);
  3. exports the built sample tuples as flat CSR arrays -> ``packed_<name>.npz`` (derived
     data, so the GPU box -- which has no /root/reference -- can run the real datasets).
No reference source text is written anywhere; only data (inputs and expected outputs).
"""
import importlib.util
import os
import pickle
import sys
import tempfile

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
DATASETS = {
    "digital_music": "Digital_Music",
    "clothing": "Clothing_Shoes_and_Jewelry",
}
# the other five datasets the reference ships (README.md:36,38-41): their review logs are exported like the
# two above; of what the real build_dataset.py builds from them only a digest is kept (sha256 per array of
# the CSR export + sizes + the first samples): tests/golden/digest_<name>.json
MORE = {
    "beauty": "Beauty",
    "home_kitchen": "Home_and_Kitchen",
    "office": "Office_Products",
    "toys": "Toys_and_Games",
    "video_games": "Video_Games",
}


def build_dataset(data_name):
    work = tempfile.mkdtemp(prefix="tlsan_fx_")
    os.makedirs(os.path.join(work, "TLSAN"))
    os.symlink(os.path.join(REF, "Data"), os.path.join(work, "Data"))
    src = open(os.path.join(REF, "TLSAN", "build_dataset.py")).read()
    assert "../Data/Digital_Music.pkl" in src
    src = src.replace("../Data/Digital_Music.pkl", "../Data/%s.pkl" % data_name)
    cwd = os.getcwd()
    os.chdir(os.path.join(work, "TLSAN"))
    try:
        exec(compile(src, "build_dataset.py", "exec"), {"__name__": "__main__"})
        with open("dataset.pkl", "rb") as f:
            train_set = pickle.load(f)
            test_set = pickle.load(f)
            counts = pickle.load(f)
            item_cate_list = pickle.load(f)
    finally:
        os.chdir(cwd)
    return train_set, test_set, counts, item_cate_list


def load_ref_input():
    spec = importlib.util.spec_from_file_location("ref_input", os.path.join(REF, "TLSAN", "input.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def pack(samples, is_test):
    """Flat CSR export of the sample tuples of build_dataset.py:58-59 (train) / :71 (test)."""
    n = len(samples)
    u = np.array([t[0] for t in samples], np.int32)
    hoff = np.zeros(n + 1, np.int64)
    soff = np.zeros(n + 1, np.int64)
    for k, t in enumerate(samples):
        assert len(t[1]) == len(t[3])
        hoff[k + 1] = hoff[k] + len(t[1])
        soff[k + 1] = soff[k] + len(t[2])
    hist = np.fromiter((x for t in samples for x in t[1]), np.int32, hoff[-1])
    # time weights are python floats 1/k, k in 1..12 (build_dataset.py:18-21); input.py:35,43,49
    # stores them into a float32 array, so float32 is lossless w.r.t. what the model sees.
    hist_t64 = np.fromiter((x for t in samples for x in t[3]), np.float64, hoff[-1])
    hist_t = hist_t64.astype(np.float32)
    sess = np.fromiter((x for t in samples for x in t[2]), np.int32, soff[-1])
    out = dict(u=u, hist_off=hoff, hist=hist, hist_t=hist_t, sess_off=soff, sess=sess)
    if is_test:
        out["pos"] = np.array([t[4][0] for t in samples], np.int32)
        out["neg"] = np.array([t[4][1] for t in samples], np.int32)
        out["cate"] = np.array([t[5] for t in samples], np.int32)
    else:
        out["target"] = np.array([t[4] for t in samples], np.int32)
        out["label"] = np.array([t[5] for t in samples], np.int8)
        out["cate"] = np.array([t[6] for t in samples], np.int32)
    return out


def record_batches(ref_input, data, cls_name, batch_size, k, which):
    it = getattr(ref_input, cls_name)(data, batch_size, k)
    rec = {}
    n_batches = it.epoch_size
    want = set(w if w >= 0 else n_batches + w for w in which)
    for step, batch in it:
        bi = step - 1
        if bi not in want:
            continue
        u, i, yj, hist_i, hist_i_new, hist_t, sl, new_sl, c = batch
        assert hist_i.dtype == np.int64 and hist_t.dtype == np.float32
        pre = "%s_bs%d_k%d_b%d_" % (cls_name, batch_size, k, bi)
        rec[pre + "u"] = np.array(u, np.int64)
        rec[pre + "i"] = np.array(i, np.int64)
        rec[pre + "yj"] = np.array(yj, np.int64)
        rec[pre + "hist_i"] = hist_i
        rec[pre + "hist_i_new"] = hist_i_new
        rec[pre + "hist_t"] = hist_t
        rec[pre + "sl"] = np.array(sl, np.int64)
        rec[pre + "new_sl"] = np.array(new_sl, np.int64)
        rec[pre + "c"] = np.array(c, np.int64)
    rec["%s_bs%d_k%d_nbatches" % (cls_name, batch_size, k)] = np.array(n_batches)
    return rec


def export_reviews():
    """The INPUT of build_dataset.py as plain arrays -> ``reviews_<name>.npz``: the remapped review
    log (reviewerID, asin, unixReviewTime in days; DataFrame row order), the item -> category map
    and the counts that ``Data/<name>.pkl`` holds.  Together with ``packed_<name>.npz`` (the tuples
    the real script built from it) this pins tlsan_amd/build_dataset.py."""
    for short, data_name in {**DATASETS, **MORE}.items():
        with open(os.path.join(REF, "Data", data_name + ".pkl"), "rb") as f:
            reviews_df, meta_df = pickle.load(f)
            icl = pickle.load(f)
            counts = pickle.load(f)
        assert (meta_df["asin"].values == np.arange(len(meta_df))).all()
        assert (meta_df["categories"].values == np.asarray(icl)).all()   # what the script's per-item lookup returns
        np.savez_compressed(
            os.path.join(OUT, "reviews_%s.npz" % short),
            reviewerID=reviews_df["reviewerID"].values.astype(np.int32), asin=reviews_df["asin"].values.astype(np.int32),
            unixReviewTime=reviews_df["unixReviewTime"].values.astype(np.int32),
            item_cate_list=np.asarray(icl, np.int32), counts=np.array(counts, np.int64))
        print("reviews_%s.npz: %d rows" % (short, len(reviews_df)))


def digest(tr, te, counts, icl):
    """sha256 of every array of the CSR export (dtypes normalised as tlsan_amd.input.PackedSet holds them)."""
    import hashlib
    norm = dict(u=np.int64, hist_off=np.int64, hist=np.int64, hist_t=np.float32, sess_off=np.int64, sess=np.int64,
                cate=np.int64, target=np.int64, label=np.int64, pos=np.int64, neg=np.int64)
    out = {"counts": [int(x) for x in counts], "item_cate_list": hashlib.sha256(np.asarray(icl, np.int32).tobytes()).hexdigest()}
    for prefix, d in (("train_", tr), ("test_", te)):
        out[prefix + "n"] = int(len(d["u"]))
        for k, v in d.items():
            out[prefix + k] = hashlib.sha256(np.ascontiguousarray(np.asarray(v).astype(norm[k])).tobytes()).hexdigest()
    out["train_head"] = {k: np.asarray(tr[k][:8]).tolist() for k in ("u", "target", "label", "cate")}
    out["test_head"] = {k: np.asarray(te[k][:8]).tolist() for k in ("u", "pos", "neg", "cate")}
    return out


def export_digests():
    import json
    for short, data_name in MORE.items():
        print("building", data_name, flush=True)
        train_set, test_set, counts, icl = build_dataset(data_name)
        print("  train %d test %d counts %s" % (len(train_set), len(test_set), counts), flush=True)
        json.dump(digest(pack(train_set, False), pack(test_set, True), counts, icl),
                  open(os.path.join(OUT, "digest_%s.json" % short), "w"), indent=1)


def export_epoch_order():
    """What the reference's training loop feeds the model: ``random.seed(1234)`` at import (train.py:15), then
    per epoch ``random.shuffle(train_set)`` and ``DataInput(train_set, 32, Ls)`` (train.py:190-192).  train.py
    itself cannot be imported (TensorFlow), so those two lines are executed here on the sample list the real
    build_dataset.py built (packed_<name>.npz) with the real input.py; recorded: the first two and the last
    batch of epochs 1 and 2 -> ``epoch_order_<name>.npz``."""
    import random
    sys.path.insert(0, os.path.dirname(os.path.dirname(OUT)))
    from tlsan_amd.build_dataset import to_samples
    from tlsan_amd.input import load_packed
    ref_input = load_ref_input()
    for short in DATASETS:
        train_set = to_samples(load_packed(os.path.join(OUT, "packed_%s.npz" % short))[0])
        random.seed(1234)                                    # train.py:15
        rec = {}
        for epoch in (1, 2):
            random.shuffle(train_set)                        # train.py:191
            r = record_batches(ref_input, train_set, "DataInput", 32, 10, [0, 1, -1])
            rec.update({"e%d_%s" % (epoch, k): v for k, v in r.items()})
        np.savez_compressed(os.path.join(OUT, "epoch_order_%s.npz" % short), **rec)
        print("epoch_order_%s.npz" % short)


def main():
    if "--reviews-only" in sys.argv:
        return export_reviews()
    if "--epoch-order" in sys.argv:
        return export_epoch_order()
    if "--digests-only" in sys.argv:
        export_reviews()
        return export_digests()
    export_reviews()
    ref_input = load_ref_input()
    for short, data_name in DATASETS.items():
        print("building", data_name, flush=True)
        train_set, test_set, counts, icl = build_dataset(data_name)
        print("  train %d test %d counts %s" % (len(train_set), len(test_set), counts))
        tr, te = pack(train_set, False), pack(test_set, True)
        np.savez_compressed(
            os.path.join(OUT, "packed_%s.npz" % short),
            counts=np.array(counts, np.int64), item_cate_list=np.asarray(icl, np.int32),
            **{"train_" + k: v for k, v in tr.items()}, **{"test_" + k: v for k, v in te.items()})
        rec = {}
        # reference defaults: train batch 32, test batch 128, Ls 10 (train.py:29,44,45)
        rec.update(record_batches(ref_input, train_set, "DataInput", 32, 10, [0, 1, 2, 3, 4, 5, 6, 7, -1]))
        rec.update(record_batches(ref_input, test_set, "DataInputTest", 128, 10, [0, 1, -1]))
        # a shorter window (more truncation) and a big batch
        rec.update(record_batches(ref_input, train_set, "DataInput", 64, 4, [0, 3, -1]))
        rec.update(record_batches(ref_input, test_set, "DataInputTest", 50, 3, [0, -1]))
        rec.update(record_batches(ref_input, train_set, "DataInput", 1024, 10, [0, -1]))
        np.savez_compressed(os.path.join(OUT, "batches_%s.npz" % short), **rec)
    export_digests()
    export_epoch_order()
    print("done")


if __name__ == "__main__":
    sys.exit(main())
