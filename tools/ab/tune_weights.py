"""Dev tool (GPU box): fixed-point tuning of K1q's per-wave shares (COOLPUPPY_AMD_K1Q_WEIGHTS) from the phase clocks.
Every iteration runs tools/k1_probe.py with the phase-clock variant, reads the mean clocks every wave spent in its window
loop, and moves each control wave's share towards equal time; then all candidates are timed un-instrumented, interleaved."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def probe(w, variant, reps):
    env = dict(os.environ, COOLPUPPY_AMD_K1Q_WEIGHTS=",".join(str(int(round(x))) for x in w))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "k1_probe.py"), "--variants", str(variant), "--reps", str(reps)],
                         capture_output=True, text=True, env=env, cwd=ROOT).stdout
    res, ph = None, None
    for line in out.splitlines():
        if line.startswith("{"):
            d = json.loads(line)
            if "k1_ms" in d:
                res = d
            if "windows_by_wave_mean" in d:
                ph = d
    return res, ph


def main():
    w = [911, 911, 911, 911, 1044, 1044, 1044, 1044, 1024, 1024, 1024, 1024, 850, 850, 850, 850]
    cands = [list(w)]
    for it in range(4):
        res, ph = probe(w, 67108864, 5)
        t = ph["windows_by_wave_mean"]
        ctrl = list(range(2, 16))
        mean = sum(t[i] for i in ctrl) / len(ctrl)
        print("iter", it, "k1_ms(timed)", res["k1_ms"], "barrier1", ph["phases_mean_clk_per_wave"]["barrier1"], "windows", [int(x / 1000) for x in t], flush=True)
        for i in ctrl:
            w[i] = w[i] * (mean / t[i]) ** 0.8
        # ROI waves: equal time between the two
        m2 = (t[0] + t[1]) / 2
        for i in (0, 1):
            w[i] = w[i] * (m2 / t[i]) ** 0.8
        cands.append(list(w))
    best = {}
    for rep in range(3):
        for k, c in enumerate(cands):
            res, _ = probe(c, 0, 9)
            best.setdefault(k, []).append(res["k1_ms"])
    for k, c in enumerate(cands):
        print("cand", k, [int(round(x)) for x in c], "k1_ms", best[k], flush=True)


main()
