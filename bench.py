#!/usr/bin/env python3
"""bench.py -- user-sequences/s of the TLSAN train step (forward + backward + update) on the
Electronics-scale synthetic workload (BASELINE.json configs[2] shapes; SURVEY.md 8d inputs).

  python bench.py [--gpus N] [--steps K] [--warmup W]
  (N > 1: one process per GPU.  Under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`
   the ranks come from the launcher's environment; started plainly, bench.py launches its N rank processes
   itself -- before this process touches the GPU -- and relays rank 0's JSON line.)

One "step" = one pass of the hot path over one batch of 4096 synthetic user-sequences that is
already resident in HBM.  Rank 0 prints ONE JSON line with the whole-job throughput, the
roofline record of the dominant kernel (k_fwd_bwd; HIP events on its own stream, in a pass of the
same steps directly behind the timed region -- the events cost the steps that carry them) and the CPU baseline (the oracle's op-for-op torch port, timed on host cores
on a bounded sample).  The oracle is only a baseline/checker here; the timed path is the HIP
library (tlsan_amd/libtlsan_hip.so) and nothing else.
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
F32_MFMA_PEAK_TFLOPS = 157.3  # same guide: peak FP32 (matrix), v_mfma_f32_16x16x4_f32, exact f32


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="electronics")
    ap.add_argument("--batch", type=int, default=0, help="override the config's batch size")
    ap.add_argument("--n-batches", type=int, default=16, help="distinct resident batches cycled through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--profile-level", type=int, default=1)
    ap.add_argument("--l2-mode", default="lazy", choices=["lazy", "dense"],
                    help="lazy: W = P*W_stored, only used rows touched (same update as the reference's dense L2); "
                         "dense: every row decayed every step")
    ap.add_argument("--table-dtype", default="f32", choices=["f32", "bf16"],
                    help="storage of the item/user/category tables (arithmetic is fp32 either way)")
    ap.add_argument("--also-bf16", type=int, default=1,
                    help="single GPU, fp32 run: also time the same steps with bf16 tables (BASELINE.json configs[2] names "
                         "bf16 storage) and report them under 'bf16_tables'")
    ap.add_argument("--graph", type=int, default=0, help="replay hipGraph-captured steps instead of eager launches (single-GPU path)")
    ap.add_argument("--prefetch", type=int, default=2,
                    help="eager mode: build the destination index of the next batch(es) on a second stream: 1 = one step ahead, 2 = two")
    ap.add_argument("--event-every", type=int, default=4,
                    help="kernel-timing pass (behind the timed region): every Nth step carries the HIP events that bracket k_fwd_bwd")
    ap.add_argument("--force-sharded", action="store_true", help="run the sharded (multi-GPU) code path even at N=1")
    ap.add_argument("--sharded-graph", type=int, default=0,
                    help="sharded path at ONE rank (--force-sharded), static-shape step: replay hipGraph-captured steps (ShardedModel.capture_step); "
                         "over several ranks the steps are always issued eagerly.  Off by default: measured SLOWER than eager steps "
                         "(104-105 vs 85 us/step at one rank, profiles/r06_sharded.md: a recorded step joins its side streams at its end)")
    ap.add_argument("--wire-dtype", default="f32", choices=["f32", "bf16"],
                    help="sharded path: rows cross the wire with fp32 or bf16 embedding values (the owners' weights stay fp32)")
    ap.add_argument("--static-rows", type=int, default=1,
                    help="sharded path, lazy L2: fixed-size row exchanges (no exchange size passes through the host)")
    ap.add_argument("--accuracy-steps", type=int, default=2000,
                    help="accuracy leg (rank 0, N=1): train steps of the reference protocol on the real Digital-Music samples, "
                         "HIP path and fp64 oracle side by side, then AUC / P@20 / R@20 of both (0: skip)")
    return ap.parse_args()


def self_launch(args):
    """--gpus N > 1 without a launcher's environment: start the N ranks as child processes (this process has not
    imported torch or touched a GPU yet; it never replaces itself), relay rank 0's JSON line, fail if any rank fails."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    text = out.decode(errors="replace")
    lines = [l for l in text.splitlines() if l.startswith("{") and '"metric"' in l]
    if any(codes) or not lines:
        sys.stderr.write(text[-4000:])
        raise SystemExit("bench.py: rank exit codes %s" % codes)
    print(lines[-1], flush=True)


def lib_id(path):
    import hashlib
    try:
        with open(path, "rb") as f:
            h = hashlib.sha256(f.read()).hexdigest()[:16]
    except OSError:
        h = None
    return {"path": os.path.relpath(path, ROOT) if path.startswith(ROOT) else path, "sha256_16": h,
            "overridden_by_TLSAN_LIB_PATH": bool(os.environ.get("TLSAN_LIB_PATH"))}


def accuracy_leg(nsteps, dev):
    """BASELINE.json's metric names accuracy beside throughput ("AUC@20": AUC, and P@20 / R@20): the
    reference's protocol (train.py:26-49: d=64, batch 32, lr 1.0, L2 5e-5, clip 5; its shuffle stream) on the real
    Digital-Music samples for `nsteps` steps, HIP path and fp64 oracle from the same initial values on the same
    batches, then test AUC (model.py:263) and P@20 / R@20 (model.py:146,153) of both.  The oracle is the checker."""
    from oracle import tlsan_oracle as orc
    from tlsan_amd import synth
    from tlsan_amd.input import DataInput, DataInputTest, load_packed
    from tlsan_amd.model import Model
    from tlsan_amd.train import epoch_rng
    path = os.path.join(ROOT, "tests", "golden", "packed_digital_music.npz")
    train_set, test_set, (U, I, C), icl = load_packed(path)
    cfg = synth.make_config("clothing", hidden_units=64, itemid_embedding_size=32, userid_embedding_size=32,
                            cateid_embedding_size=32, user_count=U, item_count=I, cate_count=C)
    m = Model(cfg, icl, device=dev)
    p = {k: v.astype(np.float64) for k, v in m.get_params().items()}
    rng, step, t_gpu, t_orc = epoch_rng(), 0, 0.0, 0.0
    while step < nsteps:
        train_set.shuffle(rng)
        for _, batch in DataInput(train_set, 32, 10):
            t0 = time.perf_counter()
            m.train_async(batch, 1.0)
            t1 = time.perf_counter()
            _, p, _ = orc.train_step(p, icl, orc.as_batch(batch), 8, cfg["regulation_rate"], 1.0)
            t_gpu, t_orc = t_gpu + t1 - t0, t_orc + time.perf_counter() - t1
            step += 1
            if step >= nsteps:
                break
    ks, n = (1, 10, 20, 30, 40, 50), len(test_set)
    g_auc = o_auc = 0.0
    o_hits = np.zeros(len(ks))
    for _, tb in DataInputTest(test_set, 128, 10):
        g_auc += m.eval_auc(None, tb) * len(tb[0])
        m.eval_prec(None, tb)
        m.eval_recall(None, tb)
        ob = orc.as_batch(tb, True)
        a, _, _ = orc.eval_auc_batch(p, icl, ob, 8)
        o_auc += a * len(tb[0])
        ut = orc.forward(p, icl, dict(ob, y=np.zeros(len(tb[0]))), 8)["u_t"]
        o_hits += np.asarray(orc.hits_at_k(orc.all_item_scores(p, icl, ut), np.asarray(tb[1]), ks), np.float64)
    gp, gr = m.prec_20.eval(), m.recall_20.eval()
    return {"dataset": "Digital-Music (real samples, tests/golden/packed_digital_music.npz)", "steps": step,
            "protocol": "reference defaults (d=64, batch 32, lr 1.0, L2 5e-5, clip 5, train.py's shuffle stream); README.md:35 "
                        "reports 0.9753 after 20 epochs (23.7 k steps)",
            "hip": {"auc": round(g_auc / n, 6), "p_at_20": round(float(gp), 6), "r_at_20": round(float(gr), 6)},
            "oracle_fp64": {"auc": round(o_auc / n, 6), "p_at_20": round(float(o_hits[2] / (20 * n)), 6),
                            "r_at_20": round(float(o_hits[2] / n), 6)},
            "auc_abs_diff": round(abs(g_auc - o_auc) / n, 9), "test_users": n}


def cpu_baseline(cfg, icl, batch, seconds):
    """oracle/tlsan_torch_ref.py: eager fp32 torch on host cores, dense L2 grads over whole
    tables, global-norm clip, SGD -- the reference's step restated op for op (kind 'port':
    TensorFlow 1.8 cannot run here or on the GPU box)."""
    import torch
    from oracle import tlsan_torch_ref as tref
    from oracle import tlsan_oracle as orc
    ncpu = os.cpu_count() or 1
    p = tref.params_to_torch(orc.init_params(cfg, seed=1234, dtype=np.float32), dtype=torch.float32)
    b = tref.batch_to_torch(orc.as_batch(batch), dtype=torch.float32)
    B = len(batch[0])
    # eager torch on many small ops does not scale with threads: probe a few counts, keep the best
    best, cores, probe = None, 1, {}
    for th in sorted({1, min(8, ncpu), min(32, ncpu)}):
        torch.set_num_threads(th)
        tref.train_step_(p, icl, b, cfg["num_heads"], cfg["regulation_rate"], 1.0)  # warm
        t1 = time.perf_counter()
        tref.train_step_(p, icl, b, cfg["num_heads"], cfg["regulation_rate"], 1.0)
        dt1 = time.perf_counter() - t1
        probe[th] = round(B / dt1, 1)
        if best is None or dt1 < best:
            best, cores = dt1, th
    torch.set_num_threads(cores)
    n, t0 = 0, time.perf_counter()
    while True:
        tref.train_step_(p, icl, b, cfg["num_heads"], cfg["regulation_rate"], 1.0)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= seconds or n >= 200:
            break
    res = dict(value=round(n * B / dt, 1), unit="user-sequences/s", cores=cores, kind="port",
               sample="%d train steps of batch %d (same synthetic batch 0, fp32 eager torch, %.1f s)" % (n, B, dt))
    # SURVEY 8d asks for these beside it (short samples, same port): one thread at the config's batch,
    # the reference's default batch 32 (train.py:44), and eval_auc's two forward passes at its test batch 128
    from tlsan_amd import synth

    def rate(fn, nseq, budget):
        fn()
        k, t1 = 0, time.perf_counter()
        while True:
            fn()
            k += 1
            e = time.perf_counter() - t1
            if e >= budget or k >= 400:
                return round(k * nseq / e, 1)
    also = {"train_1_thread_batch_%d" % B: probe.get(1)}
    b32 = tref.batch_to_torch(orc.as_batch(synth.make_batches(cfg, 1, 32, seed=99)[0]), dtype=torch.float32)
    also["train_batch_32"] = rate(lambda: tref.train_step_(p, icl, b32, cfg["num_heads"], cfg["regulation_rate"], 1.0), 32, 2.0)
    tb = orc.as_batch(synth.make_batches(cfg, 1, 128, seed=98, test=True)[0], is_test=True)
    tb_i = tref.batch_to_torch(dict(tb, y=np.zeros(128, np.float32)), dtype=torch.float32)
    tb_j = tref.batch_to_torch(dict(tb, i=tb["j"], y=np.zeros(128, np.float32)), dtype=torch.float32)

    def eval_auc():
        with torch.no_grad():
            tref.forward(p, icl, tb_i, cfg["num_heads"])
            tref.forward(p, icl, tb_j, cfg["num_heads"])
    also["eval_auc_batch_128"] = rate(eval_auc, 128, 2.0)
    res["also"] = also
    return res


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    from tlsan_amd import _lib as L
    from tlsan_amd import synth
    from tlsan_amd.model import DeviceBatch, Model

    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = "cuda:%d" % local_rank
    dist = None
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))

    cfg = synth.make_config(args.workload)
    B = args.batch or cfg["train_batch_size"]
    icl = synth.item_cate_list(cfg)
    # weak scaling: every rank trains its own batch of B sequences per step
    host_batches = synth.make_batches(cfg, args.n_batches, B, seed=1234 + 1000 * rank)
    if not sharded:
        model = Model(cfg, icl, device=dev, l2_mode=args.l2_mode, table_dtype=args.table_dtype)
        stepper = model
    else:
        from tlsan_amd.dist import ShardedModel
        # lazy L2: the static-shape step (fixed-size exchanges, nothing passes through the host)
        model = ShardedModel(cfg, icl, device=dev, l2_mode=args.l2_mode, static_rows=bool(args.static_rows) and args.l2_mode == "lazy",
                             wire_dtype=args.wire_dtype)
        stepper = model
    dbs = [stepper.device_batch(b) for b in host_batches]
    lr = 1.0
    lib = L.load()

    def committed_traffic(key):
        """HBM-side bytes per launch of k_fwd_bwd come from rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs,
        FETCH doubled per MI355X_MICROARCH.md); counters cannot be read from inside this process, so the line carries the
        committed measurement WITH its provenance (profiles/traffic.json, scripts/refresh_profiles.sh).  key: None (the
        fp32 headline) / 'bf16_tables' / 'bf16_mfma'.  Only for the bench's own shape."""
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if not (os.path.exists(tpath) and args.workload == "electronics" and B == cfg["train_batch_size"] and not sharded):
            return None, None
        try:
            tj = json.load(open(tpath))
            node = tj if key is None else tj.get(key)
            if not node or "k_fwd_bwd_hbm_bytes_per_launch" not in node:
                return None, None
            t = node["k_fwd_bwd_hbm_bytes_per_launch"]
            src = {"bytes": t, "source": "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH "
                   "doubled; not measured by this run)", "round": tj.get("round"),
                   "FETCH_SIZE_KB": node.get("FETCH_SIZE_KB"), "WRITE_SIZE_KB": node.get("WRITE_SIZE_KB")}
            if key is None and "step" in tj:
                src["step_hbm_bytes"] = tj["step"].get("hbm_bytes_per_step")
            return t, src
        except Exception:
            return None, None

    use_graph = bool(args.graph) and not sharded
    # (Geo::FUSE_DK + fused_dk() in tlsan_api.hip: the dK product rides in k_fwd_bwd -- no k_dk_partial launch -- for
    #  d <= 128 and at most 256 sample groups per launch)
    nsb = 16
    lib_fused = cfg["hidden_units"] <= 128 and (B + nsb - 1) // nsb <= 256
    graphs = [model.capture_step(db, lr) for db in dbs] if use_graph else None
    # the one-rank sharded step as recorded graphs: capture and replay alternately along the batch cycle
    # (its length is a multiple of the four plan slots), then replay in that order -- ShardedModel.capture_step
    use_graph_sh = bool(sharded and world == 1 and args.sharded_graph and model.static_rows and len(dbs) % 4 == 0)
    if use_graph_sh:
        for s in range(4):        # (one-time initialisation cannot be recorded)
            stepper.train_async(dbs[s % len(dbs)], lr, next_batch=dbs[(s + 1) % len(dbs)])
        torch.cuda.synchronize()
        graphs = []
        for k in range(len(dbs)):
            g = model.capture_step(dbs[k], dbs[(k + 1) % len(dbs)], lr)
            model.replay(g)
            graphs.append(g)
        torch.cuda.synchronize()

    host = [0.0]      # seconds the host spent inside the enqueue calls of the timed steps (is the step bound by its host thread?)

    def run(n, first, timed=False, clock=False):
        t_in, p_in = (time.perf_counter(), getattr(stepper, "poll_seconds", 0.0)) if clock else (0.0, 0.0)
        for s in range(n):
            k = (first + s) % len(dbs)
            if use_graph:
                if timed and args.profile_level > 0 and s % args.event_every == 0:
                    stepper.train_async(dbs[k], lr)
                else:
                    model.replay(graphs[k])
            elif use_graph_sh:
                model.replay(graphs[k])
            elif sharded and model.static_rows:
                # the next two batches are known (as in any input pipeline): their routing plans are built beside this step
                stepper.train_async(dbs[k], lr, next_batch=dbs[(k + 1) % len(dbs)], after_next=dbs[(k + 2) % len(dbs)])
            elif sharded:
                stepper.train_async(dbs[k], lr, next_batch=dbs[(k + 1) % len(dbs)])
            elif args.prefetch:
                # the next two batches are known (as in any input pipeline): their destination indices are built on a
                # second stream, two steps ahead (--prefetch 1: one step ahead)
                stepper.train_async(dbs[k], lr, next_batch=dbs[(k + 1) % len(dbs)],
                                    after_next=dbs[(k + 2) % len(dbs)] if args.prefetch >= 2 else None)
            else:
                stepper.train_async(dbs[k], lr)
        if clock:
            host[0] = time.perf_counter() - t_in - (getattr(stepper, "poll_seconds", 0.0) - p_in)

    def fence():
        torch.cuda.synchronize()
        if dist is not None and world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    # the destination indices are built two batches ahead: the first steps behind a fence still build their own (and
    # their successors') -- a few untimed steps beyond --warmup let that pipeline fill, so that a short timed window
    # (the round-end driver times 20 steps after 5) reads the steady state (round 3: 65.8 us/step there, 61.3 over 200)
    fill = 4 if (args.prefetch >= 2 or (sharded and model.static_rows)) and not use_graph and not use_graph_sh else 0
    run(args.warmup + fill, 0)
    fence_warm = args.warmup + fill
    if fill:          # (no fence between the fill steps and the timed ones would be better still; the contract wants one)
        torch.cuda.synchronize()
    fence()
    if hasattr(stepper, "started_at") and not use_graph:
        stepper.started_at = []      # host time stamps of the steps' kernel starts (no extra GPU work: the word is polled anyway)
    t0 = time.perf_counter()
    run(args.steps, fence_warm, timed=False, clock=True)
    fence()
    dt = time.perf_counter() - t0
    starts = np.asarray(getattr(stepper, "started_at", None) or [], np.float64)
    if hasattr(stepper, "started_at"):
        stepper.started_at = None
    if os.environ.get("BENCH_DUMP_STARTS") and rank == 0 and len(starts) > 1:     # (diagnostic: where a short window's extra time goes)
        print("window %.1f us; first start +%.1f us after t0, last start %.1f us before the end; start-to-start (us): %s" % (
            dt * 1e6, (starts[0] - t0) * 1e6, (t0 + dt - starts[-1]) * 1e6, " ".join("%.0f" % (x * 1e6) for x in np.diff(starts))), file=sys.stderr)
    # the kernel's own duration: a pair of HIP events ATTACHED TO k_fwd_bwd's dispatch on its stream (hipExtLaunchKernelGGL
    # through tlsan_profile_*: the dispatch's own begin / end time stamps, which is what a rocprofv3 kernel trace reports),
    # live, in a pass of the SAME steps right behind the timed ones.  (Rounds 2-3 recorded two events AROUND the launch:
    # barrier packets of their own that read ~3 us longer than the trace and cost the step that carried them ~6 us.)
    nprof = min(args.steps, 4096)
    lib.tlsan_profile_stride(1 if use_graph else args.event_every)   # graph mode: only the eager steps reach the marks
    lib.tlsan_profile_enable(args.profile_level)
    run(nprof, fence_warm + args.steps, timed=True)     # (continues the batch cycle: the next two batches are announced)
    fence()
    buf = (ctypes.c_float * (nprof * 5))()
    nrec = lib.tlsan_profile_collect(buf, nprof)
    lib.tlsan_profile_enable(0)
    seg = np.frombuffer(buf, dtype=np.float32)[: nrec * 5].reshape(nrec, 5) if nrec > 0 else np.zeros((0, 5))
    if dist is not None and world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = float(model.last_loss.item()) if sharded else float(model._out[0].item())
    if sharded and model.static_rows:
        model.check_static_overflow()     # a batch that did not fit the fixed exchange voids the run: raise, print nothing
    static_rows = bool(sharded and model.static_rows)
    also = also_mm = None
    if args.also_bf16 and not sharded and args.table_dtype == "f32" and not use_graph:
        # same batches, same step count with (a) the tables stored as bf16 -- the storage BASELINE.json configs[2] names --
        # and fp32 arithmetic, (b) bf16 tables AND bf16 matrix products (v_mfma_f32_16x16x16_bf16: operands rounded to
        # bf16, fp32 products and sums; logits then agree with the fp32 oracle to ~1e-3 of their scale, not 1e-4).
        # Reported beside the fp32 headline, never instead of it.
        def variant(table_dtype, matrix_dtype, what, tkey):
            import gc
            gc.collect()                  # (the previous model's buffers and streams go before the next one is built)
            torch.cuda.synchronize()
            mv = Model(cfg, icl, device=dev, l2_mode=args.l2_mode, table_dtype=table_dtype, matrix_dtype=matrix_dtype)
            for s in range(fence_warm):
                mv.train_async(dbs[s % len(dbs)], lr, next_batch=dbs[(s + 1) % len(dbs)], after_next=dbs[(s + 2) % len(dbs)])
            torch.cuda.synchronize()
            mv.started_at = []
            t1 = time.perf_counter()
            for s in range(args.steps):
                k = (fence_warm + s) % len(dbs)
                mv.train_async(dbs[k], lr, next_batch=dbs[(k + 1) % len(dbs)], after_next=dbs[(k + 2) % len(dbs)])
            torch.cuda.synchronize()
            dtv = time.perf_counter() - t1
            sv, mv.started_at = np.asarray(mv.started_at, np.float64), None
            spread = {"ms_per_step_p%d" % q: round(float(np.percentile(np.diff(sv), q)) * 1e3, 4) for q in (10, 50, 90)} if len(sv) >= 3 else {}
            lib.tlsan_profile_stride(args.event_every)      # (kernel time: a pass of its own, as for the headline)
            lib.tlsan_profile_enable(args.profile_level)
            for s in range(args.steps):
                k = (fence_warm + args.steps + s) % len(dbs)
                mv.train_async(dbs[k], lr, next_batch=dbs[(k + 1) % len(dbs)], after_next=dbs[(k + 2) % len(dbs)])
            torch.cuda.synchronize()
            pb = (ctypes.c_float * (nprof * 5))()
            nr = lib.tlsan_profile_collect(pb, nprof)
            lib.tlsan_profile_enable(0)
            kms = float(np.frombuffer(pb, dtype=np.float32)[: nr * 5].reshape(nr, 5)[:, 1].mean()) if nr > 0 else float("nan")
            abv = [synth.algorithmic_bytes(cfg, host_batches[(fence_warm + s) % len(host_batches)], 2) for s in range(args.steps)]
            step_b, k_b = float(np.mean([x["train_step"] for x in abv])), float(np.mean([x["fwd_bwd_kernel"] for x in abv]))
            ach = k_b / (kms * 1e-3) / 1e9
            return {"value": round(args.steps * B / dtv, 1), "unit": "user-sequences/s", "ms_per_step": round(dtv / args.steps * 1e3, 4), **spread,
                    "what": what, "step_algorithmic_bytes": round(step_b),
                    "step_frac": round(step_b / (dtv / args.steps) / 1e9 / HBM_PEAK_GBS, 4),
                    "roofline": {"bound": "hbm", "kernel": "k_fwd_bwd", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                 "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": committed_traffic(tkey)[0],
                                 "traffic_source": committed_traffic(tkey)[1], "algorithmic_bytes_per_launch": round(k_b),
                                 "kernel_ms": round(kms, 5)},
                    "final_loss": round(float(mv._out[0].item()), 6)}
        # one model at a time: with the measured model still alive the variants' steps were occasionally 1.3-4x longer
        # (same kernel times; 2 of 6 runs), without it never in 10
        model = stepper = None
        also = variant("bf16", "f32", "item/user/category tables bf16, fp32 arithmetic, stochastic rounding on update", "bf16_tables")
        also_mm = variant("bf16", "bf16", "bf16 tables + bf16 matrix products (operands rounded to bf16, fp32 accumulate); "
                                          "parity: tests/test_gpu_parity.py::test_bf16_matrix_products", "bf16_mfma")
    if not np.isfinite(loss):
        raise SystemExit("bench.py: non-finite loss %r" % loss)

    if rank == 0:
        seqs = args.steps * B * world
        eb = 2 if (args.table_dtype == "bf16" and not sharded) else 4   # e = bytes per table element
        ab = [synth.algorithmic_bytes(cfg, host_batches[(fence_warm + s) % len(host_batches)], eb) for s in range(args.steps)]
        k_bytes = float(np.mean([a["fwd_bwd_kernel"] for a in ab]))
        l2 = args.l2_mode
        step_bytes = float(np.mean([a["train_step"] for a in ab])) + (synth.dense_sweep_bytes(cfg) if l2 == "dense" else 0)
        k_flops = float(np.mean([synth.algorithmic_flops(cfg, host_batches[(fence_warm + s) % len(host_batches)]) for s in range(args.steps)]))
        k_ms = float(seg[:, 1].mean()) if nrec else float("nan")
        achieved = k_bytes / (k_ms * 1e-3) / 1e9 if nrec else None
        traffic, traffic_src = committed_traffic(None) if eb == 4 else (None, None)
        out = {
            "metric": "user-sequences/sec (train step: fwd+bwd+update), Electronics-scale",
            "value": round(seqs / dt, 1),
            "unit": "user-sequences/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "pipeline_fill_steps": fill,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            # spread of the timed window's steps: intervals between consecutive steps' kernel starts as the host saw them
            # (the pinned `started` word train_async polls anyway: no event, no extra launch).  K - 1 intervals of K steps;
            # `value` is still whole-window sequences / wall time (fences included)
            **({"ms_per_step_p10": round(float(np.percentile(np.diff(starts), 10)) * 1e3, 4),
                "ms_per_step_p50": round(float(np.percentile(np.diff(starts), 50)) * 1e3, 4),
                "ms_per_step_p90": round(float(np.percentile(np.diff(starts), 90)) * 1e3, 4)} if len(starts) >= 3 else {}),
            # (time the host needed to ENQUEUE the timed steps, per step, without its waits for the GPU -- the poll of the
            #  `started` word --: well below ms_per_step when the GPU is the bound)
            "host_enqueue_ms_per_step": round(host[0] / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "%s-scale synthetic (U=%d I=%d C=%d, d=%d, Ls=%d), batch %d/GPU, %s tables, "
                                   "l2_mode=%s"
                                   % (args.workload, cfg["user_count"], cfg["item_count"], cfg["cate_count"],
                                      cfg["hidden_units"], cfg["Ls"], B,
                                      "bf16 (fp32 arithmetic, stochastic rounding on update)" if eb == 2 else "fp32",
                                      "dense (every row decayed every step, as the reference)" if l2 == "dense" else
                                      "lazy (reference's dense-L2 update as W = P*W_stored; only used rows touched)"),
                       "global_batch": B * world, "parallelism": "1 process/GPU, tables %s"
                       % ("on one GPU" if not sharded else "row-sharded (id %% N), RCCL all-to-all + one all-reduce" +
                          (", static-shape exchanges, plans two batches ahead" if static_rows else "") +
                          (", bf16 rows on the wire (fp32 weights at the owners)" if args.wire_dtype == "bf16" else ""))},
            "roofline": {"bound": "hbm", "kernel": "k_fwd_bwd", "achieved": None if achieved is None else round(achieved, 1),
                         "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": round(k_bytes), "kernel_ms": round(k_ms, 5),
                         "kernel_ms_how": "HIP events attached to the kernel's dispatch (hipExtLaunchKernelGGL): its own begin / end time "
                                          "stamps, as in a rocprofv3 kernel trace; every %dth step of a pass behind the timed steps" % args.event_every,
                         "step_algorithmic_bytes": round(step_bytes),
                         "step_frac": round(step_bytes / (dt / args.steps) / 1e9 / HBM_PEAK_GBS, 4)},
            # the designated roofline is HBM (north_star); at d=128 in exact fp32 the same kernel's maps
            # are also 1.4 GFLOP on the fp32 matrix pipe (SURVEY 8d "compute co-bound"): reported beside it
            "roofline_mfma": {"bound": "mfma", "kernel": "k_fwd_bwd", "dtype": "f32",
                              "achieved": None if not nrec else round(k_flops / (k_ms * 1e-3) / 1e12, 2),
                              "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": None if not nrec else round(k_flops / (k_ms * 1e-3) / 1e12 / F32_MFMA_PEAK_TFLOPS, 4),
                              "algorithmic_flops_per_launch": round(k_flops)},
            "launch": ("hipGraph replay (1 graph/step; every %dth step eager for the HIP-event kernel timing)" % args.event_every)
                      if use_graph else "hipGraph replay of the recorded static-shape sharded step (1 graph/step, the next batch's plan inside)" if use_graph_sh else ("eager, %d launches/step on the main stream + the destination index (2 launches) of the "
                                         "batch %s on a second stream" % (3 if lib_fused else 4, "after next" if args.prefetch >= 2 else "next")
                                         if (args.prefetch and not sharded) else "eager"),
            "static_overflow_checked": True if (sharded and static_rows) else None,
            # which build was measured (TLSAN_LIB_PATH can point the loader at a diagnostic build: it would show here)
            "library": lib_id(L.LIB_PATH),
            "final_loss": round(loss, 6),
        }
        if also is not None:
            out["bf16_tables"] = also
        if also_mm is not None:
            out["bf16_mfma"] = also_mm
        if nrec and args.profile_level >= 2:
            out["segments_ms"] = {n: round(float(seg[:, i].mean()), 5) for i, n in enumerate(L.PROF_SEGMENTS)}
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(cfg, icl, host_batches[0], args.cpu_seconds)
        if args.accuracy_steps > 0 and world == 1 and not sharded:
            out["accuracy"] = accuracy_leg(args.accuracy_steps, dev)
        result = json.dumps(out)
    else:
        result = None
    if dist is not None:
        dist.destroy_process_group()
    if result is not None:
        # RCCL prints its version banner through C stdio; flush that first so the JSON line is
        # the last line on stdout
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(result, flush=True)


if __name__ == "__main__":
    main()
