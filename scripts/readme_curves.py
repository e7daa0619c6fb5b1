#!/usr/bin/env python3
"""From the JSON files of scripts/readme_band.py (one per regulation_rate): per dataset the test-AUC-vs-step curve
(mean over the initialisations), the step of its maximum, and how far it falls from there to the end of the run.
    python scripts/readme_curves.py name=file.json [name=file.json ...] [--golden out.json]"""
import json, sys
import numpy as np
args = [a for a in sys.argv[1:] if "=" in a and not a.startswith("--")]
golden = sys.argv[sys.argv.index("--golden") + 1] if "--golden" in sys.argv else None
sets = {a.split("=", 1)[0]: json.load(open(a.split("=", 1)[1])) for a in args}
gold = {}
for tag, rows in sets.items():
    print("\n### regulation_rate = %s\n" % tag)
    print("| dataset | steps | README | best AUC (mean ± σ) | at step (share of the run) | AUC at the end | fall from the best | inside README ± 2σ* |")
    print("|---|---|---|---|---|---|---|---|")
    inside = 0
    for r in rows:
        curves = [np.array(x["curve"], float) for x in r["runs"]]
        n = min(len(c) for c in curves)
        steps = curves[0][:n, 0]
        mean = np.mean([c[:n, 1] for c in curves], 0)
        k = int(mean.argmax())
        best = np.array([x["best_auc"] for x in r["runs"]])
        sg = float(best.std(ddof=1)) if len(best) > 1 else float("nan")
        band = 2 * float(np.hypot(sg, r["sampling_sigma"]))       # run-to-run and test-set sampling, both
        ok = abs(best.mean() - r["readme_auc"]) <= band
        inside += ok
        total = r["runs"][0]["steps"]
        print("| %s | %d | %.4f | %.4f ± %.4f | %d (%.0f %%) | %.4f | %+.4f | %s |" % (
            r["readme_name"], total, r["readme_auc"], best.mean(), sg, steps[k], 100.0 * steps[k] / total, mean[-1], mean[-1] - mean[k], "yes" if ok else "no"))
        gold.setdefault(r["dataset"], {})[tag] = dict(mean=round(float(best.mean()), 5), sigma=round(sg, 5), n=len(best),
                                                       readme=r["readme_auc"], sampling_sigma=round(r["sampling_sigma"], 5),
                                                       best_step=int(steps[k]), steps=int(total))
    print("\n%d of %d datasets inside README ± 2σ* (σ* = run-to-run σ and the test set's sampling σ in quadrature)" % (inside, len(rows)))
    print("\nCurves (mean test AUC over the initialisations at every evaluation, step:AUC):\n")
    for r in rows:
        curves = [np.array(x["curve"], float) for x in r["runs"]]
        n = min(len(c) for c in curves)
        mean = np.mean([c[:n, 1] for c in curves], 0)
        st = curves[0][:n, 0]
        sel = sorted(set(list(range(0, n, max(1, n // 12))) + [int(mean.argmax()), n - 1]))
        print("* %s: " % r["readme_name"] + "  ".join("%dk:%.4f" % (st[i] / 1000, mean[i]) for i in sel))
if golden:
    json.dump(gold, open(golden, "w"), indent=1, sort_keys=True)
