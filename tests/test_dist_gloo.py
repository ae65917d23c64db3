"""world_size-2 CPU test (gloo) of the multi-GPU path: LPT sharding of region calls + all-reduce of the packed
accumulators + finaliser must reproduce the single-process (and the reference's golden) result."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import torch.distributed as dist
import golden_util as gu
from coolpuppy_amd import coolpup, dist as pdist
from oracle import pileup_oracle as po

dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)

def run_plan_sharded(pu, plan):
    """CPU stand-in for PileUpper.run_plan: this rank's share on the oracle, then the real all-reduce."""
    indptr, col, cnt = pu._aclr.pixel_table()
    bins = pu.clr.bins()
    weight = bins[plan["weight_name"]][:].values if plan["weight_name"] else None
    cov = bins[plan["cov_name"]][:].values if plan["cov_name"] else None
    acc = po.empty_acc(plan["T"], plan["pad"])
    rank, world = pdist.world()
    mine = []
    for c in plan["calls"]:
        full = len(c["r0"])
        c = pdist.slice_call(c, rank, world)
        mine.append((len(c["r0"]), full))
        # the engine would rebuild tiles from tile_ptr: check the slice's tile_ptr agrees with its tile array
        assert np.array_equal(np.concatenate([[0], np.cumsum(np.bincount(c["tile"], minlength=plan["T"]))]), c["tile_ptr"])
        if c["flip_from"] is not None:
            fl = c["flip"].astype(bool)
            for t in range(plan["T"]):
                seg = fl[c["tile_ptr"][t]:c["tile_ptr"][t + 1]]
                k = int(c["flip_from"][t] - c["tile_ptr"][t])
                assert not seg[:k].any() and seg[k:].all()
        if len(c["r0"]):
            for expected, sc in coolpup.iter_expected_subcalls(plan, c):
                po.pileup_c(indptr, col, cnt, weight, cov, expected, sc["r0"], sc["c0"], sc["flip"], sc["tile"],
                            plan["T"], plan["pad"], sc["ignore_diags"], sc["mode"], acc=acc)
    T, W = plan["T"], 2 * plan["pad"] + 1
    f64 = np.concatenate([acc["sum"].ravel(), acc["cov_start"].ravel(), acc["cov_end"].ravel()])
    i64 = np.concatenate([acc["num"].ravel(), acc["n"].ravel()])
    f64, i64 = pdist.allreduce_arrays(f64, i64)
    acc["sum"] = f64[:T*W*W].reshape(T, W, W); acc["cov_start"] = f64[T*W*W:T*W*W+T*W].reshape(T, W)
    acc["cov_end"] = f64[T*W*W+T*W:].reshape(T, W)
    acc["num"] = i64[:T*W*W].reshape(T, W, W); acc["n"] = i64[T*W*W:]
    return acc, mine

shares = {{}}
def patched(pu, plan):
    acc, mine = run_plan_sharded(pu, plan)
    shares[len(shares)] = (mine, len(plan["calls"]))
    return acc
coolpup.PileUpper.run_plan = patched
for name in {names!r}:
    z, df = gu.run(name, coolpup.pileup)
    gu.compare(z, df, rtol=1e-12)
rank, world = pdist.world()
assert world == 2
print("RANK", rank, "OK", json.dumps({{k: v for k, v in shares.items()}}))
dist.destroy_process_group()
'''


def test_two_rank_gloo_matches_golden(tmp_path, oracle_mod):
    names = ["G3_nshifts3", "G6c_by_strand_distance_controls", "G4b_expected_not_ooe", "G7_trans_bedpe_expected",
             "G2b_raw_covnorm_controls"]
    port = 29500 + (os.getpid() % 2000)
    script = tmp_path / "worker.py"
    script.write_text(WORKER.format(root=ROOT, port=port, names=names))
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                              text=True) for r in range(2)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n{out[-3000:]}"
        assert f"RANK {r} OK" in out
    # the two ranks' shares of every call add up to the call and are nearly equal
    import json
    s0 = json.loads(outs[0].split("OK", 1)[1])
    s1 = json.loads(outs[1].split("OK", 1)[1])
    for k in s0:
        for (a, full), (b, _) in zip(s0[k][0], s1[k][0]):
            assert a + b == full and abs(a - b) <= max(2, full // 50 + 200)


def test_shard_is_deterministic_and_balanced():
    from coolpuppy_amd import dist as pdist
    w = [100, 1, 50, 49, 3, 97]
    parts = [pdist.shard(len(w), weights=w, rank=r, world_size=3) for r in range(3)]
    assert sorted(i for p in parts for i in p) == list(range(len(w)))
    loads = [sum(w[i] for i in p) for p in parts]
    assert max(loads) - min(loads) <= 4
    assert parts == [pdist.shard(len(w), weights=w, rank=r, world_size=3) for r in range(3)]


GPU_WORKER = r'''
import os, sys, json
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
os.environ["COOLPUPPY_AMD_DEVICE"] = "0"
import torch.distributed as dist
import golden_util as gu
from coolpuppy_amd import coolpup
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size=2)
out = {{}}
for name in {names!r}:
    z, df = gu.run(name, coolpup.pileup)
    gu.compare(z, df, rtol=1e-6)
    out[name] = int(len(df))
dist.barrier()
if dist.get_rank() == 0:
    print("RESULT " + json.dumps(out))
dist.destroy_process_group()
'''


@pytest.mark.gpu
def test_two_ranks_share_one_gpu_and_match_goldens(hip_lib, tmp_path):
    """The library's multi-GPU path on real hardware: two processes (both on GPU 0, gloo rendezvous) each pile up
    their slice of every engine call, exchange the packed accumulators (pup_export -> all-reduce -> pup_import) and
    must both end with the reference's golden result."""
    names = ["G1_bedpe_balanced", "G3_nshifts3", "G6c_by_strand_distance_controls", "G4_expected_ooe", "G2_raw_covnorm"]
    names = [n for n in names if os.path.exists(os.path.join(ROOT, "tests", "golden", n + ".npz"))]
    assert len(names) >= 3
    script = tmp_path / "gpu_worker.py"
    script.write_text(GPU_WORKER.format(root=ROOT, port=29517, names=names))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, str(script), str(r)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              text=True, env=env) for r in range(2)]
    outs = [p.communicate(timeout=600) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, se[-3000:]
    assert any("RESULT" in so for so, _ in outs)
