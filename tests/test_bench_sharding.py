"""bench.py's multi-GPU arithmetic: the chromosome sharding (a pure function every rank evaluates on its own) on CPU, and
— on the GPU box — a two-rank run of bench.py itself (gloo, both ranks on GPU 0) against the one-rank run of the same
small workload: same window counts and the same centre enrichment after the all-reduce."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_shard_snippets_partitions_the_workload():
    import bench
    rng = np.random.default_rng(3)
    co = np.array([0, 900, 1500, 1800, 2600, 2650, 4000], np.int64)
    n, n_roi = 50_000, 4_000
    r0 = rng.integers(0, 3990, n).astype(np.int32)
    c0 = (r0 + rng.integers(0, 10, n)).astype(np.int32)
    whole = bench.shard_snippets(r0, c0, n_roi, co, 0, 1)
    assert whole[2].tolist() == [0, n_roi, n] and whole[3] == [(0, 4000)]
    for world in (2, 3, 8):
        parts = [bench.shard_snippets(r0, c0, n_roi, co, rank, world) for rank in range(world)]
        owners = [p[4] for p in parts]
        assert all(np.array_equal(o, owners[0]) for o in owners)              # every rank computes the same assignment
        # the ranks' windows partition the workload, tile by tile, in the original order
        chrom = np.searchsorted(co, r0, side="right") - 1
        for rank, (pr0, pc0, tp, rows, owner) in enumerate(parts):
            mine = owner[chrom] == rank
            np.testing.assert_array_equal(pr0, r0[mine])
            np.testing.assert_array_equal(pc0, c0[mine])
            assert tp[0] == 0 and tp[1] == int(mine[:n_roi].sum()) and tp[2] == int(mine.sum())
            # every window of the rank lies in a row range the rank holds
            lo = np.array([a for a, _ in rows]); hi = np.array([b for _, b in rows])
            if len(pr0):
                k = np.searchsorted(lo, pr0, side="right") - 1
                assert (k >= 0).all() and (pr0 < hi[k]).all()
        assert sum(int(p[2][-1]) for p in parts) == n
        rows_all = sorted(r for p in parts for r in p[3])
        assert rows_all == [(int(co[k]), int(co[k + 1])) for k in range(len(co) - 1)]     # disjoint, complete
        # longest-processing-time balance: no rank carries more than the biggest chromosome above the mean
        loads = np.array([int(p[2][-1]) for p in parts], float)
        assert loads.max() <= n / world + np.bincount(chrom).max()


def _bench(args, env=None, nproc=1, port=29561):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py")] if nproc == 1 else \
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
         "--master-port", str(port), os.path.join(ROOT, "bench.py")]
    res = subprocess.run(cmd + args, capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-3000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_two_ranks_on_one_gpu_match_one_rank(hip_lib, tmp_path):
    common = ["--steps", "3", "--warmup", "1", "--pairs", "30000", "--chroms", "5", "--lam", "400", "--cpu-sample", "0"]
    env = {"TMPDIR": str(tmp_path)}
    one = _bench(["--gpus", "1"] + common, env)
    two = _bench(["--gpus", "2", "--backend", "gloo", "--scaling", "strong"] + common, dict(env, COOLPUPPY_AMD_BENCH_DEVICE="0"), nproc=2)
    assert one["n_gpus"] == 1 and one["scaling"] == "strong" and two["n_gpus"] == 2 and two["scaling"] == "strong"
    assert two["check"]["n"] == one["check"]["n"]                               # all-reduced tiles = the single-GPU tiles
    assert abs(two["check"]["center_roi_over_ctrl"] / one["check"]["center_roi_over_ctrl"] - 1) < 1e-12
    assert two["config"]["snippets_per_step"] == one["config"]["snippets_per_step"] == sum(one["check"]["n"])
    assert two["strong"]["value"] == two["value"]
    w = two["weak"]
    assert w["scaling"] == "weak" and w["pairs"] == 60000 and sum(w["check_n"]) == w["snippets_per_step"]
    assert w["snippets_per_step"] > 1.9 * one["config"]["snippets_per_step"]
    assert two["exchange"].startswith("torch.distributed")                     # gloo: no RCCL between ranks sharing a GPU
    # the default N > 1 line: weak scaling (per-GPU work as at N = 1) as `value`, the strong reading beside it
    dflt = _bench(["--gpus", "2", "--backend", "gloo"] + common, dict(env, COOLPUPPY_AMD_BENCH_DEVICE="0"), nproc=2, port=29563)
    assert dflt["scaling"] == "weak" and dflt["value"] == dflt["weak"]["value"] and dflt["steps"] == 3
    assert dflt["config"]["pairs"] == 60000 and dflt["strong"]["snippets_per_step"] == one["config"]["snippets_per_step"]


@pytest.mark.gpu
@pytest.mark.parametrize("config", [3, 4])
def test_bench_config_3_and_4_two_ranks_on_one_gpu_match_one_rank(hip_lib, tmp_path, config):
    """`bench.py --config 3` (grouped: chromosomes sharded) and `--config 4` (trans: region PAIRS sharded)
    through the library's plan path with two gloo ranks on GPU 0 — the all-reduced tiles must hold exactly the windows of the
    one-rank run, tile by tile."""
    common = ["--config", str(config), "--steps", "2", "--warmup", "1", "--pairs", "30000", "--chroms", "5", "--lam", "400",
              "--cpu-sample", "0", "--trans-nnz", "400000"]
    env = {"TMPDIR": str(tmp_path)}
    one = _bench(["--gpus", "1"] + common, env)
    two = _bench(["--gpus", "2", "--backend", "gloo"] + common, dict(env, COOLPUPPY_AMD_BENCH_DEVICE="0"), nproc=2, port=29571 + config)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2
    assert two["check"]["n"] == one["check"]["n"] and sum(one["check"]["n"]) > 20000
    assert two["config"]["snippets_per_step"] == one["config"]["snippets_per_step"] == one["check"]["n_sum"]
    assert one["config"]["tiles"] == two["config"]["tiles"] and (one["config"]["tiles"] >= 30 if config == 3 else one["config"]["tiles"] <= 2)
    assert one["roofline"]["kernel_family"] and one["cpu_baseline"] is None


def test_exchange_verdict_labels_a_fallback_and_refuses_when_strict():
    """bench.py --gpus N with --exchange native: a communicator spanning the job -> the engine's own all-reduce under its name; a
    missing or smaller one -> the torch fallback, LABELLED and flagged (round 6), or — --strict-exchange — no line at all."""
    import bench
    ok = bench.exchange_verdict(8, 8, False)
    assert ok == {"fallback": False, "message": "", "exchange": "pup_allreduce (RCCL on the engine's stream, in place)"}
    assert bench.exchange_verdict(8, 8, True) == ok
    for spans in (None, 1, 4):
        v = bench.exchange_verdict(8, spans, False, rank=3)
        assert v["fallback"] and v["exchange"].startswith("FALLBACK: torch.distributed.all_reduce") and f"{spans} of 8" in v["exchange"]
        assert "rank 3" in v["message"] and "FALLING BACK" in v["message"]
        with pytest.raises(SystemExit, match="refusing to report"):
            bench.exchange_verdict(8, spans, True)
    a = bench.parse(["--gpus", "8"])
    assert a.exchange == "native" and a.strict_exchange is False
    assert bench.parse(["--gpus", "8", "--strict-exchange"]).strict_exchange is True
