"""Batcher: counterpart of the reference's ``TLSAN/input.py`` (DataInput :4-54,
DataInputTest :57-107).

Same iteration protocol and the same 9-tuple per batch,
``(u, i, y|j, hist_i, hist_i_new, hist_t, sl, new_sl, c)``, with the same padding rules:
``sl = min(len(hist), k)``; when the history is longer than ``k`` the LAST ``k`` items are
kept (input.py:41-45), otherwise it is left-aligned (:47-49); ``hist_i_new`` is padded to
the longest session of the batch (:33,37,50-51); the last batch may be short (:22).

The reference walks python lists sample by sample (about 150 k samples/s/core); here the
sample tuples are packed once into flat CSR arrays (:class:`PackedSet`) and each batch is
three vectorised gathers.  Outputs are bit-identical to the reference's on the committed
fixtures (tests/test_input_parity.py).  Scalars per sample come back as int64 numpy arrays
instead of python lists (same values; ``len(batch[0])`` etc. keep working).
"""
from __future__ import annotations

import numpy as np


class PackedSet:
    """Flat (CSR) form of a list of sample tuples.

    train tuple (build_dataset.py:58-59): (user, hist[], session[], hist_t[], target, label, cate)
    test tuple  (build_dataset.py:71):    (user, hist[], session[], hist_t[], (pos, neg), cate)
    """

    def __init__(self, u, hist_off, hist, hist_t, sess_off, sess, cate, target=None, label=None,
                 pos=None, neg=None):
        self.u = np.asarray(u, np.int64)
        self.hist_off = np.asarray(hist_off, np.int64)
        self.hist = np.asarray(hist, np.int64)
        self.hist_t = np.asarray(hist_t, np.float32)
        self.sess_off = np.asarray(sess_off, np.int64)
        self.sess = np.asarray(sess, np.int64)
        self.cate = np.asarray(cate, np.int64)
        self.is_test = pos is not None
        if self.is_test:
            self.pos = np.asarray(pos, np.int64)
            self.neg = np.asarray(neg, np.int64)
        else:
            self.target = np.asarray(target, np.int64)
            self.label = np.asarray(label, np.int64)
        self.order = np.arange(len(self.u), dtype=np.int64)

    def __len__(self):
        return len(self.order)

    # -- constructors -------------------------------------------------------------------
    @classmethod
    def from_samples(cls, samples):
        """From the python list of tuples that ``dataset.pkl`` holds (train.py:131-135)."""
        n = len(samples)
        if n == 0:
            raise ValueError("empty sample list")
        is_test = isinstance(samples[0][4], (tuple, list))
        hoff = np.zeros(n + 1, np.int64)
        soff = np.zeros(n + 1, np.int64)
        hoff[1:] = np.cumsum([len(t[1]) for t in samples])
        soff[1:] = np.cumsum([len(t[2]) for t in samples])
        hist = np.fromiter((x for t in samples for x in t[1]), np.int64, hoff[-1])
        hist_t = np.fromiter((x for t in samples for x in t[3]), np.float64, hoff[-1]).astype(np.float32)
        sess = np.fromiter((x for t in samples for x in t[2]), np.int64, soff[-1])
        u = [t[0] for t in samples]
        if is_test:
            return cls(u, hoff, hist, hist_t, soff, sess, [t[5] for t in samples],
                       pos=[t[4][0] for t in samples], neg=[t[4][1] for t in samples])
        return cls(u, hoff, hist, hist_t, soff, sess, [t[6] for t in samples],
                   target=[t[4] for t in samples], label=[t[5] for t in samples])

    @classmethod
    def from_npz(cls, z, prefix):
        g = lambda k: z[prefix + k]
        if prefix + "pos" in z:
            return cls(g("u"), g("hist_off"), g("hist"), g("hist_t"), g("sess_off"), g("sess"),
                       g("cate"), pos=g("pos"), neg=g("neg"))
        return cls(g("u"), g("hist_off"), g("hist"), g("hist_t"), g("sess_off"), g("sess"),
                   g("cate"), target=g("target"), label=g("label"))

    def shuffle(self, rng):
        """Epoch shuffle (train.py:191 shuffles the python list in place; here the index order, in place,
        so that consecutive epochs compose as the reference's do).  `rng`: a `random.Random` (the
        reference's stream, CPython's Fisher-Yates over a list) or a numpy RandomState / Generator."""
        import random as _random
        if isinstance(rng, _random.Random):
            o = self.order.tolist()
            rng.shuffle(o)
            self.order = np.asarray(o, np.int64)
        else:
            rng.shuffle(self.order)

    # -- the batch ----------------------------------------------------------------------
    def make_batch(self, lo, hi, k):
        idx = self.order[lo:hi]
        hlo, hhi = self.hist_off[idx], self.hist_off[idx + 1]
        length = hhi - hlo
        sl = np.minimum(length, k)                                   # input.py:31
        start = hlo + np.maximum(length - k, 0)                      # last k when longer (:41-45)
        ar = np.arange(k, dtype=np.int64)[None, :]
        valid = ar < sl[:, None]
        src = np.where(valid, start[:, None] + ar, 0)
        nh = len(self.hist)
        if nh == 0:
            hist_i = np.zeros((len(idx), k), np.int64)
            hist_t = np.zeros((len(idx), k), np.float32)
        else:
            hist_i = np.where(valid, self.hist[src], 0).astype(np.int64)
            hist_t = np.where(valid, self.hist_t[src], np.float32(0)).astype(np.float32)
        slo, shi = self.sess_off[idx], self.sess_off[idx + 1]
        new_sl = shi - slo                                           # input.py:32
        max_new = int(new_sl.max()) if len(idx) else 0               # :33
        ar2 = np.arange(max_new, dtype=np.int64)[None, :]
        valid2 = ar2 < new_sl[:, None]
        src2 = np.where(valid2, slo[:, None] + ar2, 0)
        if max_new == 0 or len(self.sess) == 0:
            hist_i_new = np.zeros((len(idx), max_new), np.int64)
        else:
            hist_i_new = np.where(valid2, self.sess[src2], 0).astype(np.int64)
        u = self.u[idx]
        c = self.cate[idx]
        if self.is_test:
            return (u, self.pos[idx], self.neg[idx], hist_i, hist_i_new, hist_t, sl, new_sl, c)
        return (u, self.target[idx], self.label[idx], hist_i, hist_i_new, hist_t, sl, new_sl, c)


class _Input:
    def __init__(self, data, batch_size, k):
        self.k = k
        self.batch_size = batch_size
        self.data = data if isinstance(data, PackedSet) else PackedSet.from_samples(data)
        n = len(self.data)
        self.epoch_size = n // batch_size + (1 if n % batch_size else 0)   # input.py:9-11
        self.i = 0

    def __iter__(self):
        return self

    def __next__(self):
        if self.i == self.epoch_size:
            raise StopIteration
        lo = self.i * self.batch_size
        hi = min((self.i + 1) * self.batch_size, len(self.data))
        self.i += 1
        return self.i, self.data.make_batch(lo, hi, self.k)


class DataInput(_Input):
    """Training batches: (u, i, y, hist_i, hist_i_new, hist_t, sl, new_sl, c) (input.py:54)."""

    def __init__(self, data, batch_size, k):
        super().__init__(data, batch_size, k)
        if self.data.is_test:
            raise ValueError("DataInput needs train tuples")


class DataInputTest(_Input):
    """Test batches: (u, i_pos, j_neg, hist_i, hist_i_new, hist_t, sl, new_sl, c) (input.py:107)."""

    def __init__(self, data, batch_size, k):
        super().__init__(data, batch_size, k)
        if not self.data.is_test:
            raise ValueError("DataInputTest needs test tuples")


def load_packed(path):
    """Load a ``packed_<name>.npz`` export: (train PackedSet, test PackedSet, (U,I,C), item_cate_list)."""
    z = np.load(path)
    counts = tuple(int(x) for x in z["counts"])
    return (PackedSet.from_npz(z, "train_"), PackedSet.from_npz(z, "test_"), counts,
            z["item_cate_list"].astype(np.int32))
