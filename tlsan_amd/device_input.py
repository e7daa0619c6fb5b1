"""Device-resident batcher: ``DataInput`` / ``DataInputTest`` of the reference
(TLSAN/input.py:4-54, 57-107) with the sample set kept in HBM as CSR arrays and every batch
assembled by one HIP launch (``tlsan_batch_pack``): no per-batch host work beyond choosing the
padded session length, no host -> device copy.  Same iteration protocol (``for i, batch in ...``);
the batch is a :class:`tlsan_amd.model.DeviceBatch`, which ``Model.train`` / ``eval_auc`` /
``ShardedModel`` accept directly; ``batch.to_host()`` gives the reference's numpy 9-tuple
(bit-identical to ``tlsan_amd.input`` and to the reference on the committed fixtures).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib as L
from .input import PackedSet
from .model import DeviceBatch


class DevicePackedSet:
    """A :class:`PackedSet` uploaded once; ``shuffle`` permutes the sample order of the next epoch."""

    def __init__(self, ps: PackedSet, device="cuda:0"):
        self.ps = ps
        self.device = torch.device(device)
        self.is_test = ps.is_test
        t = lambda a, dt=torch.int32: torch.as_tensor(np.ascontiguousarray(a)).to(dt).to(self.device)
        if max(len(ps.hist), len(ps.sess)) >= 2 ** 31:
            raise ValueError("sample set too large for int32 offsets")
        self.u, self.cate = t(ps.u), t(ps.cate)
        self.hist_off, self.sess_off = t(ps.hist_off), t(ps.sess_off)
        # (one pad element so an empty history / session array still has an address)
        self.hist = t(np.concatenate([ps.hist, [0]]))
        self.hist_t = t(np.concatenate([ps.hist_t, np.zeros(1, np.float32)]), torch.float32)
        self.sess = t(np.concatenate([ps.sess, [0]]))
        self.target = t(ps.pos if ps.is_test else ps.target)
        self.second = t(ps.neg if ps.is_test else ps.label)
        self._sess_len = (ps.sess_off[1:] - ps.sess_off[:-1]).astype(np.int64)   # host copy: picks Sn per batch
        self.c = L.Packed(len(ps.u), *(x.data_ptr() for x in (self.u, self.cate, self.hist_off, self.hist, self.hist_t,
                                                               self.sess_off, self.sess, self.target, self.second)))
        self.set_order(ps.order)

    def __len__(self):
        return len(self.ps)

    def set_order(self, order):
        self._order_host = np.asarray(order, np.int64)
        self.order = torch.as_tensor(self._order_host.astype(np.int32)).to(self.device)

    def shuffle(self, rng):
        """Epoch shuffle (train.py:191), same stream as PackedSet.shuffle."""
        self.ps.shuffle(rng)
        self.set_order(self.ps.order)

    def make_batch(self, lo, hi, k):
        B = hi - lo
        Sn = int(self._sess_len[self._order_host[lo:hi]].max()) if B > 0 else 0   # input.py:33
        db = DeviceBatch.allocate(B, Sn, k, self.device, self.is_test)
        lib = L.load()
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        L.check(lib.tlsan_batch_pack(C.byref(self.c), self.order.data_ptr(), lo, C.byref(db.c), k, 1 if self.is_test else 0, st),
                "tlsan_batch_pack")
        return db


class _DeviceInput:
    def __init__(self, data, batch_size, k, device="cuda:0"):
        self.k, self.batch_size = k, batch_size
        self.data = data if isinstance(data, DevicePackedSet) else DevicePackedSet(
            data if isinstance(data, PackedSet) else PackedSet.from_samples(data), device)
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


class DeviceDataInput(_DeviceInput):
    """Training batches (input.py:54) assembled on the device."""

    def __init__(self, data, batch_size, k, device="cuda:0"):
        super().__init__(data, batch_size, k, device)
        if self.data.is_test:
            raise ValueError("DeviceDataInput needs train tuples")


class DeviceDataInputTest(_DeviceInput):
    """Test batches (input.py:107) assembled on the device."""

    def __init__(self, data, batch_size, k, device="cuda:0"):
        super().__init__(data, batch_size, k, device)
        if not self.data.is_test:
            raise ValueError("DeviceDataInputTest needs test tuples")
