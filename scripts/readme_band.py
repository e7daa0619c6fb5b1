#!/usr/bin/env python3
"""The reference's README AUC table (README.md:29-41) on the seven datasets it ships, through the HIP path.

For every dataset: the review log (tests/golden/reviews_<name>.npz, exported from Data/<name>.pkl) is turned
into samples by tlsan_amd.build_dataset (bit-identical to the reference's build_dataset.py: tests/
test_build_dataset.py), then tlsan_amd.train runs the reference's protocol (train.py:26-49 defaults: d=64,
batch 32, lr 1.0, L2 5e-5, clip 5, 20 epochs, evaluation every 1000 steps, the epoch shuffle stream of
train.py:15,191) and reports what the reference prints as "Best test_auc" (train.py:228-230,240: the
maximum over the evaluations).  TensorFlow's initial values cannot be reproduced, so every dataset is run
with several draws of the same initial distributions (--seeds) -> mean +- sigma next to the README number.

    python scripts/readme_band.py [--seeds 1234,1,2] [--datasets clothing,...] [--epochs 20] > profiles/rNN_readme_band.md
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

README = {  # README.md:34-41 (TLSAN column); the other three datasets are not shipped
    "clothing": ("Clothing-Shoes", 0.9363), "digital_music": ("Digital-Music", 0.9753), "office": ("Office-Products", 0.9773),
    "beauty": ("Beauty", 0.9368), "home_kitchen": ("Home-Kitchen", 0.8950), "video_games": ("Video-Games", 0.9459),
    "toys": ("Toys-Games", 0.9309),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", default="1234,1,2")
    ap.add_argument("--datasets", default=",".join(README))
    ap.add_argument("--epochs", type=int, default=20)
    ap.add_argument("--l2_mode", default="dense")
    ap.add_argument("--json", default=None)
    ap.add_argument("--extra", default="", help="further tlsan_amd.train flags, e.g. '--matrix_dtype bf16 --hidden_units 128'")
    a = ap.parse_args()
    from tlsan_amd import train as T
    from tlsan_amd.build_dataset import build_packed
    seeds = [int(x) for x in a.seeds.split(",")]
    rows, t_all = [], time.time()
    for name in a.datasets.split(","):
        z = np.load(os.path.join(ROOT, "tests", "golden", "reviews_%s.npz" % name))
        U, I, C = (int(x) for x in z["counts"][:3])
        t0 = time.time()
        train_set, test_set = build_packed(z["reviewerID"], z["asin"], z["unixReviewTime"], z["item_cate_list"], I)
        t_build = time.time() - t0
        res = []
        for seed in seeds:
            args = T.parse(["--quiet", "--eval_topk", "0", "--max_epochs", str(a.epochs), "--seed", str(seed),
                            "--l2_mode", a.l2_mode, "--model_dir", "/tmp/tlsan_band/%s_%d" % (name, seed)] + a.extra.split())
            train_set.order = np.arange(len(train_set.u), dtype=np.int64)      # every run starts from the built order
            r = T.train(args, data=(train_set, test_set, (U, I, C), z["item_cate_list"].astype(np.int32)))
            res.append(dict(seed=seed, init_auc=r["init_auc"], best_auc=r["best_auc"], final_auc=r["final_auc"],
                            steps=r["steps"], seconds=round(r["seconds"], 1),
                            best_step=max(r["history"], key=lambda h: h[2])[0] if r["history"] else None,
                            curve=[[int(h[0]), round(float(h[2]), 5)] for h in r["history"]]))   # (step, test AUC) at every evaluation
            print("# %s seed %d: best %.4f (step %s) final %.4f init %.4f, %d steps in %.0f s"
                  % (name, seed, r["best_auc"], res[-1]["best_step"], r["final_auc"], r["init_auc"], r["steps"], r["seconds"]),
                  file=sys.stderr, flush=True)
        best = np.array([x["best_auc"] for x in res])
        rows.append(dict(dataset=name, readme_name=README[name][0], readme_auc=README[name][1], users=U, items=I, cates=C,
                         train_samples=len(train_set), test_users=len(test_set), build_seconds=round(t_build, 2),
                         mean=float(best.mean()), sigma=float(best.std(ddof=1)) if len(best) > 1 else None,
                         sampling_sigma=float(np.sqrt(README[name][1] * (1 - README[name][1]) / len(test_set))), runs=res))
    print("| dataset | users / items / cats | train samples | README AUC | ours: best AUC, mean ± σ over %d inits | Δ | Δ/σ | test-set sampling σ |" % len(seeds))
    print("|---|---|---|---|---|---|---|---|")
    for r in rows:
        d = r["mean"] - r["readme_auc"]
        sg = r["sigma"] or float("nan")
        print("| %s | %d / %d / %d | %d | %.4f | %.4f ± %.4f (%s) | %+.4f | %+.1f | %.4f |"
              % (r["readme_name"], r["users"], r["items"], r["cates"], r["train_samples"], r["readme_auc"], r["mean"], sg,
                 ", ".join("%.4f" % x["best_auc"] for x in r["runs"]), d, d / sg if sg else float("nan"), r["sampling_sigma"]))
    print("\n(%d runs, %.0f s wall; protocol: tlsan_amd.train defaults = TLSAN/train.py:26-49; l2_mode=%s%s)"
          % (len(rows) * len(seeds), time.time() - t_all, a.l2_mode, "; extra flags: " + a.extra if a.extra else ""))
    if a.json:
        json.dump(rows, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()
