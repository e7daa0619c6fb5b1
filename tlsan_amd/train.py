"""Driver: counterpart of the reference's ``TLSAN/train.py`` (flags :26-54, loop :121-249).

    python -m tlsan_amd.train --dataset tests/golden/packed_clothing.npz [--flag=value ...]

Same flag names and defaults as the reference (``tf.app.flags`` -> argparse), same flow:
initial AUC / P@k / R@k, ``max_epochs`` epochs of shuffled batches, evaluation every
``eval_freq`` steps, learning rate 1.0 -> 0.1 at step 150000, save when the test AUC improves
(> 0.8).  ``--dataset`` takes either a ``dataset.pkl`` written by the reference's
``build_dataset.py`` (train.py:131-135) or a ``packed_<name>.npz`` export (tests/golden/).
"""
from __future__ import annotations

import argparse
import glob
import json
import os
import pickle
import random
import re
import shutil
import time

import numpy as np

from .input import DataInput, DataInputTest, PackedSet, load_packed
from .model import KS, Model

FLAGS = [  # (name, type, default)  -- train.py:26-54
    ("hidden_units", int, 64), ("num_blocks", int, 1), ("num_heads", int, 8), ("Ls", int, 10),
    ("dropout", float, 0.0), ("regulation_rate", float, 0.00005),
    ("itemid_embedding_size", int, 32), ("userid_embedding_size", int, 32), ("cateid_embedding_size", int, 32),
    ("model_dir", str, "save_path"), ("optimizer", str, "sgd"), ("learning_rate", float, 1.0),
    ("max_gradient_norm", float, 5.0), ("train_batch_size", int, 32), ("test_batch_size", int, 128),
    ("max_epochs", int, 20), ("display_freq", int, 100), ("eval_freq", int, 1000),
]


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    for name, typ, default in FLAGS:
        ap.add_argument("--" + name, type=typ, default=default)
    ap.add_argument("--from_scratch", type=lambda s: s.lower() != "false", default=True)
    ap.add_argument("--dataset", default=None, help="dataset.pkl of the reference's build_dataset.py, or a packed_<name>.npz export")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--max_steps", type=int, default=0, help="stop early (0 = run max_epochs)")
    ap.add_argument("--norm_mode", default="tf18", choices=["tf18", "dedup"])
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--table_dtype", default="f32", choices=["f32", "bf16"],
                    help="storage of item/user/category tables (bf16: fp32 arithmetic, stochastic rounding on update)")
    ap.add_argument("--wire_dtype", default="f32", choices=["f32", "bf16"],
                    help="sharded driver with --static_rows: embedding values of the rows that cross the wire (owners keep fp32)")
    ap.add_argument("--static_rows", type=int, default=0,
                    help="sharded driver, lazy L2: 0 = exchange sizes follow the batch (read by the host once a step); "
                         "1 = fixed-size exchanges sized from the first batch (x1.5); N > 1 = N row slots per rank pair. "
                         "A batch that does not fit raises at the next check (every 1024 steps, at evaluation, at the end)")
    ap.add_argument("--matrix_dtype", default="f32", choices=["f32", "bf16"],
                    help="arithmetic of the fused kernel's matrix products (bf16: operands rounded to bfloat16, fp32 accumulate)")
    ap.add_argument("--l2_mode", default="dense", choices=["dense", "lazy"])
    ap.add_argument("--eval_topk", type=int, default=1,
                    help="P@k / R@k at every evaluation point like the reference (train.py:209-218); 0: once at the end")
    ap.add_argument("--sharded", type=int, default=0,
                    help="1: row-sharded tables over the ranks of torch.distributed (implied by WORLD_SIZE > 1)")
    ap.add_argument("--seed", type=int, default=1234,
                    help="seed of the variables' initial values (the reference seeds everything with 1234, train.py:15-17; "
                         "TensorFlow's own stream cannot be reproduced, so this picks ONE draw of the same distributions)")
    ap.add_argument("--shuffle_seed", type=int, default=1234, help="seed of the epoch shuffle stream (train.py:15,191: 1234)")
    ap.add_argument("--device_input", type=int, default=1,
                    help="1: keep the sample sets in HBM and assemble batches on the device (tlsan_amd.device_input); "
                         "0: the host batcher (tlsan_amd.input), one upload per batch")
    return ap.parse_args(argv)


def load_dataset(path):
    if path.endswith(".npz"):
        return load_packed(path)
    with open(path, "rb") as f:  # train.py:131-135
        train_set = pickle.load(f)
        test_set = pickle.load(f)
        counts = pickle.load(f)
        icl = pickle.load(f)
    return PackedSet.from_samples(train_set), PackedSet.from_samples(test_set), tuple(counts), np.asarray(icl, np.int32)


def prepare_model_dir(model_dir, from_scratch, rank=0, barrier=None):
    """train.py:124-127: `from_scratch` wipes and recreates model_dir (rank 0 does it; the others wait at
    `barrier`).  Returns the checkpoint to resume from, or None -- train.py:71-76 reloads
    tf.train.get_checkpoint_state(model_dir).model_checkpoint_path, i.e. the LATEST save, when
    from_scratch is false; here that is the TLSAN-<step> with the largest step (single-file `.npz` of
    Model.save, or the `.replicated.npz` prefix of ShardedModel.save)."""
    if from_scratch:
        if rank == 0:
            if os.path.isdir(model_dir):
                shutil.rmtree(model_dir)
            os.makedirs(model_dir, exist_ok=True)
        if barrier is not None:
            barrier()
        return None
    best = None
    for f in glob.glob(os.path.join(model_dir, "TLSAN-*.npz")):
        m = re.match(r"TLSAN-(\d+)(\.replicated)?\.npz$", os.path.basename(f))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), f[:-len(".replicated.npz")] if m.group(2) else f)
    if rank == 0:
        os.makedirs(model_dir, exist_ok=True)
    if barrier is not None:
        barrier()
    return None if best is None else best[1]


def epoch_rng(seed=1234):
    """The reference's shuffle stream: `random.seed(1234)` at import (train.py:15) and
    `random.shuffle(train_set)` once per epoch (:191) -- nothing else draws from `random` in between
    (model.py and input.py never touch it), so a private `random.Random(1234)` shuffling the sample
    ORDER in place epoch after epoch visits the samples exactly as the reference's list shuffle does."""
    return random.Random(seed)


# Evaluation launches: the reference feeds its test set in batches of test_batch_size (128) and aggregates per batch
# (train.py:86-118).  What a test row contributes -- whether its pair is ranked right, the rank of its label -- does
# not depend on the rows it shares a batch with (padded session slots are masked), so the kernels run on chunks of
# EVAL_CHUNK rows (the all-items ranking reaches 39 % of the fp32 matrix peak at 4096 rows per launch and 6 % at 128)
# and the reference's per-batch aggregation is formed from slices of the per-row results: same numbers, same order.
EVAL_CHUNK = 4096


def _test_batches(test_set, config, batch_size=None):
    bs = batch_size or config["test_batch_size"]
    if isinstance(test_set, PackedSet):
        return DataInputTest(test_set, bs, config["Ls"])
    from .device_input import DeviceDataInputTest
    return DeviceDataInputTest(test_set, bs, config["Ls"])


def _per_row(model, test_set, config, fn):
    """fn(batch) -> per-row device tensor, over the whole test set in chunks of EVAL_CHUNK rows; one host copy."""
    import torch
    chunk = max(EVAL_CHUNK, config["test_batch_size"]) // config["test_batch_size"] * config["test_batch_size"]
    parts = [fn(batch) for _, batch in _test_batches(test_set, config, chunk)]
    return torch.cat(parts).cpu().numpy()


def eval_auc(model, test_set, config):
    """train.py:86-96: batch AUCs weighted by batch length."""
    ok = _per_row(model, test_set, config, model.pairs_ranked_right)
    s, bs = 0.0, config["test_batch_size"]
    for lo in range(0, len(ok), bs):
        part = ok[lo:lo + bs]
        auc_b = float(np.float32(part.sum()) / np.float32(len(part)))      # model.py:263: a float32 mean of 0/1
        s += auc_b * len(part)
    res = s / len(test_set)
    model.eval_writer.add_summary(("AUC", res), global_step=model.global_step.eval())     # train.py:91-94
    return res


def eval_prec_recall(model, test_set, config):
    """train.py:98-118: one pass for precision, one for recall (cumulative counters, as the reference); the label
    ranks both passes need are computed once."""
    ranks = _per_row(model, test_set, config, model.label_ranks)
    bs = config["test_batch_size"]
    for lo in range(0, len(ranks), bs):
        model.eval_prec(None, None, ranks=ranks[lo:lo + bs])
    prec = [getattr(model, "prec_%d" % k).eval() for k in KS]
    for lo in range(0, len(ranks), bs):
        model.eval_recall(None, None, ranks=ranks[lo:lo + bs])
    recall = [getattr(model, "recall_%d" % k).eval() for k in KS]
    step = model.global_step.eval()
    model.eval_writer.add_summary([("P@%d" % k, v) for k, v in zip(KS, prec)] +                 # train.py:103-106
                                  [("R@%d" % k, v) for k, v in zip(KS, recall)], global_step=step)   # :114-117
    return prec, recall


def train(args, data=None):
    """data (optional): (train PackedSet, test PackedSet, (U, I, C), item_cate_list) already in memory
    (tlsan_amd.build_dataset.build_packed) instead of --dataset."""
    say = (lambda *a, **k: None) if args.quiet else print
    train_set, test_set, (U, I, Cc), icl = load_dataset(args.dataset) if data is None else data
    config = {name: getattr(args, name) for name, _, _ in FLAGS}
    config.update(user_count=U, item_count=I, cate_count=Cc, from_scratch=args.from_scratch, quiet=bool(args.quiet))
    say(json.dumps(config, indent=4), flush=True)
    resume = prepare_model_dir(args.model_dir, args.from_scratch)      # train.py:124-127
    model = Model(config, icl, device=args.device, seed=args.seed, norm_mode=args.norm_mode, l2_mode=args.l2_mode,
                  table_dtype=args.table_dtype, matrix_dtype=args.matrix_dtype)
    if resume is not None:                                             # train.py:71-76 (create_model)
        say("Reloading model parameters..", flush=True)
        model.restore(None, resume)
    else:
        say("Created new model parameters..", flush=True)
    if args.device_input:
        from .device_input import DeviceDataInput, DevicePackedSet
        train_set, test_set = DevicePackedSet(train_set, args.device), DevicePackedSet(test_set, args.device)
        train_batches = lambda: DeviceDataInput(train_set, args.train_batch_size, config["Ls"])
    else:
        train_batches = lambda: DataInput(train_set, args.train_batch_size, config["Ls"])
    t0 = time.time()
    init_auc = eval_auc(model, test_set, config)
    say("Init AUC: %.4f" % init_auc)
    lr = args.learning_rate
    rng = epoch_rng(args.shuffle_seed)  # train.py:15,191
    best_auc, history = 0.0, []
    best_prec, best_recall = [0.0] * 6, [0.0] * 6              # train.py:187-188
    prec, recall = [0.0] * 6, [0.0] * 6
    import torch
    loss_sum = torch.zeros((), dtype=torch.float32, device=args.device)
    done = False
    for _ in range(args.max_epochs):
        train_set.shuffle(rng)  # train.py:191
        for batch, nxt, nxt2 in _lookahead2(model.device_batch(b) for _, b in train_batches()):
            # the reference reads the loss back every step (model.py:229-234); the sum is all the driver
            # uses, so it is accumulated on the device and read at the evaluation points only
            left = (args.max_steps - model.global_step.eval() - 1) if args.max_steps else 2     # steps after this one
            model.train_async(batch, lr, next_batch=nxt if (nxt is not None and left >= 1) else None,
                              after_next=nxt2 if (nxt2 is not None and left >= 2) else None)
            loss_sum += model._out[0]
            step = model.global_step.eval()
            if args.display_freq and step % args.display_freq == 0:    # train.py:194-195 (add_summary)
                model.train_writer.add_summary(model.train_summary(), global_step=step)
            if step % args.eval_freq == 0:
                auc = eval_auc(model, test_set, config)
                history.append((step, time.time() - t0, auc))
                say("Epoch %d Global_step %d\tTrain_loss: %.4f\tEval_auc: %.4f" %
                    (model.global_epoch_step.eval(), step, float(loss_sum.item()) / args.eval_freq, auc), flush=True)
                loss_sum.zero_()
                if args.eval_topk:                             # train.py:209-218: P@k / R@k at every evaluation
                    prec, recall = eval_prec_recall(model, test_set, config)
                    say("Precision:\n" + " ".join("@%d = %.4f" % (k, v) for k, v in zip(KS, prec)))
                    say("Recall:\n" + " ".join("@%d = %.4f" % (k, v) for k, v in zip(KS, recall)))
                    if step > 20000:                           # :222-227
                        best_prec = [max(a_, b_) for a_, b_ in zip(best_prec, prec)]
                        best_recall = [max(a_, b_) for a_, b_ in zip(best_recall, recall)]
                if auc > 0.8 and auc > best_auc:  # train.py:228-230
                    best_auc = auc
                    model.save(None)
                best_auc = max(best_auc, auc)
            if step == 150000:  # train.py:232-233
                lr = 0.1
            if args.max_steps and step >= args.max_steps:
                done = True
                break
        say("Epoch %d DONE\tCost time: %.2f" % (model.global_epoch_step.eval(), time.time() - t0), flush=True)  # :235-237
        model.global_epoch_step_op.eval()
        if done:
            break
    if not args.eval_topk or not history:   # (the reference reports what its evaluations saw; make sure there is one)
        prec, recall = eval_prec_recall(model, test_set, config)
    final_auc = eval_auc(model, test_set, config)
    best_auc = max(best_auc, final_auc)
    model.save(None)                                           # train.py:239
    model.train_writer.flush()
    model.eval_writer.flush()
    say("Best test_auc:", best_auc)
    say("Best precision:\n" + " ".join("@%d = %.4f" % (k, v) for k, v in zip(KS, best_prec)))   # :241-248
    say("Best recall:\n" + " ".join("@%d = %.4f" % (k, v) for k, v in zip(KS, best_recall)))
    say("Finished", flush=True)
    return dict(init_auc=init_auc, best_auc=best_auc, final_auc=final_auc, steps=model.global_step.eval(),
                seconds=time.time() - t0, history=history, prec=prec, recall=recall,
                best_prec=best_prec, best_recall=best_recall)


def _lookahead2(it):
    """(item, next_or_None, one_after_or_None) triples (ShardedModel.train_async(next_batch=, after_next=))."""
    buf = []
    for x in it:
        buf.append(x)
        if len(buf) == 3:
            yield buf[0], buf[1], buf[2]
            buf.pop(0)
    while buf:
        yield buf[0], (buf[1] if len(buf) > 1 else None), None
        buf.pop(0)


def _share(batch, rank, world):
    """This rank's rows of a global batch (contiguous, as even as possible) and how many of them are
    real: a rank without rows gets row 0 as a placeholder (0 real rows), so that every rank takes
    part in every collective."""
    n = len(batch[0])
    lo, hi = n * rank // world, n * (rank + 1) // world
    if hi == lo:
        return tuple(np.asarray(a)[:1] for a in batch), 0
    return tuple(np.asarray(a)[lo:hi] for a in batch), hi - lo


def _equal_share(batch, rank, world):
    """Like _share, padded (by repeating the last row) to ceil(n / world) rows: the evaluation's
    all-gather is equal-sized.  -> (rows, real rows)"""
    part, real = _share(batch, rank, world)
    want = -(-len(batch[0]) // world)
    have = len(part[0])
    if have < want:
        part = tuple(np.concatenate([a, np.repeat(a[-1:], want - have, axis=0)], 0) for a in part)
    return part, real


def train_sharded(args):
    """The same flow on N GPUs (`python -m torch.distributed.run --nproc-per-node N -m tlsan_amd.train
    --sharded 1 ...`): tables row-sharded over the ranks (tlsan_amd.dist.ShardedModel), every global
    batch of train_batch_size samples split over the ranks -- one SGD step per global batch, exactly
    the single-GPU trajectory up to fp32 summation order -- evaluation over the split test batches."""
    import torch
    import torch.distributed as dist
    from .dist import ShardedModel
    if not dist.is_initialized():
        import os
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        rank_, world_ = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", "0"))
        args.device = "cuda:%d" % local
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=rank_, world_size=world_, device_id=torch.device(args.device))
    rank, world = dist.get_rank(), dist.get_world_size()
    say = print if (rank == 0 and not args.quiet) else (lambda *a, **k: None)
    train_set, test_set, (U, I, Cc), icl = load_dataset(args.dataset)
    config = {name: getattr(args, name) for name, _, _ in FLAGS}
    config.update(user_count=U, item_count=I, cate_count=Cc, from_scratch=args.from_scratch, quiet=bool(args.quiet))
    # flags this path does not implement are refused, not ignored
    if args.table_dtype != "f32":
        raise NotImplementedError("--table_dtype %s: the sharded step keeps fp32 rows" % args.table_dtype)
    if args.norm_mode != "tf18":
        raise NotImplementedError("--norm_mode %s: the sharded step forms the clip norm as TF 1.8 does (tf18)" % args.norm_mode)
    if args.matrix_dtype != "f32":
        raise NotImplementedError("--matrix_dtype %s: the sharded step computes in fp32" % args.matrix_dtype)
    resume = prepare_model_dir(args.model_dir, args.from_scratch, rank,
                               (lambda: dist.barrier()) if world > 1 else None)     # train.py:124-127
    l2_mode = args.l2_mode if args.optimizer == "sgd" else "dense"
    if args.static_rows and l2_mode != "lazy":
        raise NotImplementedError("--static_rows is the lazy-L2 SGD step's form (--l2_mode lazy --optimizer sgd)")
    model = ShardedModel(config, icl, device=args.device, seed=args.seed, l2_mode=l2_mode,
                         static_rows=(True if args.static_rows == 1 else args.static_rows), wire_dtype=args.wire_dtype)
    if resume is not None:                                                          # train.py:71-76
        say("Reloading model parameters..", flush=True)
        model.restore(None, resume)
    dev = model.device

    def reduce_sum(vals):
        t = torch.tensor(vals, dtype=torch.float64, device=dev)
        if world > 1:
            from .dist import allreduce_sum
            allreduce_sum(t)
        return t.tolist()

    def eval_auc_():
        # train.py:86-96: sum_b auc_b * len_b / N == (pairs ranked right) / N, summed over the ranks' shares
        model.check_static_overflow()      # (static_rows: a step that did not fit its exchange is reported here at the latest)
        right = 0.0
        for _, batch in DataInputTest(test_set, config["test_batch_size"], config["Ls"]):
            part, real = _share(batch, rank, world)
            a = model.eval_auc(None, part)            # (a collective inside: every rank calls it, placeholder or not)
            right += a * real if real else 0.0
        res = reduce_sum([right])[0] / len(test_set)
        model.eval_writer.add_summary(("AUC", res), global_step=model.global_step.eval())
        return res

    def eval_pr_():
        prec = recall = None
        for _, batch in DataInputTest(test_set, config["test_batch_size"], config["Ls"]):
            part, real = _equal_share(batch, rank, world)
            prec = model.eval_prec(None, part, n_valid=real)
        for _, batch in DataInputTest(test_set, config["test_batch_size"], config["Ls"]):
            part, real = _equal_share(batch, rank, world)
            recall = model.eval_recall(None, part, n_valid=real)
        return [float(x) for x in prec], [float(x) for x in recall]

    t0 = time.time()
    init_auc = eval_auc_()
    say("Init AUC: %.4f" % init_auc)
    lr = args.learning_rate
    rng = epoch_rng(args.shuffle_seed)             # train.py:15,191; the same shuffle on every rank
    best_auc, history = 0.0, []
    best_prec, best_recall = [0.0] * 6, [0.0] * 6
    prec, recall = [0.0] * 6, [0.0] * 6
    loss_sum = torch.zeros((), dtype=torch.float32, device=dev)
    done = False
    for _ in range(args.max_epochs):
        train_set.shuffle(rng)
        shares = (_share(b, rank, world) + (len(b[0]),) for _, b in DataInput(train_set, args.train_batch_size, config["Ls"]))
        for (part, real, n_glob), nxt, nxt2 in _lookahead2((model.device_batch(p_), r_, n_) for p_, r_, n_ in shares):
            left = (args.max_steps - model.global_step.eval() - 1) if args.max_steps else 2     # steps after this one
            model.train_async(part, lr, next_batch=nxt[0] if (nxt is not None and left >= 1) else None,
                              after_next=nxt2[0] if (nxt2 is not None and left >= 2 and args.static_rows) else None,
                              weight=real * world / n_glob, sample0=n_glob * rank // world)
            loss_sum += model.last_loss[0]
            step = model.global_step.eval()
            if step % args.eval_freq == 0:
                auc = eval_auc_()
                history.append((step, time.time() - t0, auc))
                say("Epoch %d Global_step %d\tTrain_loss: %.4f\tEval_auc: %.4f" %
                    (model.global_epoch_step.eval(), step, float(loss_sum.item()) / args.eval_freq, auc), flush=True)
                loss_sum.zero_()
                if args.eval_topk:
                    prec, recall = eval_pr_()
                    say("Precision:\n" + " ".join("@%d = %.4f" % (k, v) for k, v in zip(KS, prec)))
                    say("Recall:\n" + " ".join("@%d = %.4f" % (k, v) for k, v in zip(KS, recall)))
                    if step > 20000:
                        best_prec = [max(a_, b_) for a_, b_ in zip(best_prec, prec)]
                        best_recall = [max(a_, b_) for a_, b_ in zip(best_recall, recall)]
                if auc > 0.8 and auc > best_auc:
                    best_auc = auc
                    model.save(None)
                best_auc = max(best_auc, auc)
            if step == 150000:
                lr = 0.1
            if args.max_steps and step >= args.max_steps:
                done = True
                break
        say("Epoch %d DONE\tCost time: %.2f" % (model.global_epoch_step.eval(), time.time() - t0), flush=True)
        model._epoch += 1
        if done:
            break
    if not args.eval_topk or not history:
        prec, recall = eval_pr_()
    final_auc = eval_auc_()
    best_auc = max(best_auc, final_auc)
    model.save(None)
    model.train_writer.flush()
    model.eval_writer.flush()
    say("Best test_auc:", best_auc)
    say("Finished", flush=True)
    return dict(init_auc=init_auc, best_auc=best_auc, final_auc=final_auc, steps=model.global_step.eval(),
                seconds=time.time() - t0, history=history, prec=prec, recall=recall,
                best_prec=best_prec, best_recall=best_recall, world=world)


def main(argv=None):
    import os
    args = parse(argv)
    if args.dataset is None:
        raise SystemExit("--dataset is required")
    sharded = args.sharded or int(os.environ.get("WORLD_SIZE", "1")) > 1
    res = train_sharded(args) if sharded else train(args)
    if not sharded or int(os.environ.get("RANK", "0")) == 0:
        print(json.dumps({k: v for k, v in res.items() if k != "history"}))


if __name__ == "__main__":
    main()
