"""CPU, world size 2 on gloo: the parts of the multi-GPU path that are not kernels.

* chromosome sharding (LPT) -- every chromosome owned once, makespans as SURVEY.md Appendix D;
* the percentile select loop with its only collective (histogram sum, candidate min/max):
  two ranks each histogram their own chromosomes (here with a numpy stand-in for the device
  kernel -- the host logic, rank formula and bucket walk are the product's) and must both
  arrive at the oracle's order statistics.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GENOME = [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973, 145138636,
          138394717, 133797422, 135086622, 133275309, 114364328, 107043718, 101991189, 90338345,
          83257441, 80373285, 58617616, 64444167, 46709983, 50818468, 156040895, 57227415]


def test_lpt_sharding_matches_survey_makespans():
    import genodsp_amd as gd
    want = {1: 3088269832, 2: 1546288544, 4: 776170262, 8: 400055715}
    for n, makespan in want.items():
        shards = gd.lpt_shards(GENOME, n)
        assert sorted(i for s in shards for i in s) == list(range(len(GENOME)))
        assert max(sum(GENOME[i] for i in s) for s in shards) == makespan


def test_bench_genome_is_the_survey_genome():
    import bench
    assert [n for _, n in bench.GENOME] == GENOME and sum(GENOME) == 3088269832


def test_bench_pieces_at_the_contract_rank_counts():
    """bench.py's two splits at 1/2/4/8 ranks: whole chromosomes dealt longest-first (every chromosome once, nobody
    empty) and equal stretches of the concatenated genome (shares within one base, pieces in order, every base once)"""
    import bench
    import genodsp_amd as gd
    for world in (1, 2, 4, 8):
        whole = bench.shard_pieces(GENOME, world, "chromosomes", gd.lpt_shards)
        assert sorted(c for sh in whole for c, _, _, _ in sh) == list(range(24)) and all(sh for sh in whole)
        cut = bench.shard_pieces(GENOME, world, "bases", gd.lpt_shards)
        shares = [sum(b - a for _, _, a, b in sh) for sh in cut]
        assert sum(shares) == sum(GENOME) and max(shares) - min(shares) <= 1
        seen = [0] * 24
        for sh in cut:
            for c, _, a, b in sh:
                assert a == seen[c] and a < b <= GENOME[c]
                seen[c] = b
                lo, hi = bench.piece_extent((c, 0, a, b), GENOME)
                assert lo == max(0, a - 50) and hi == min(GENOME[c], b + 50)
        assert seen == GENOME


def _np_histogram(vecs, window, lo, hi, shift, bits, prefix):
    """numpy stand-in for gdsp_select_histogram (same key image, same filter)."""
    nb = 1 << bits
    out = np.zeros(nb + 2, np.uint64)
    out[nb] = np.uint64(0xFFFFFFFFFFFFFFFF)
    for v in vecs:
        x = v[::window]
        x = x[~(x < lo) & ~(x > hi)]
        u = x.view(np.uint64).copy()
        u[u == np.uint64(0x8000000000000000)] = 0
        neg = (u >> np.uint64(63)) == 1
        key = np.where(neg, ~u, u | np.uint64(0x8000000000000000))
        if shift + bits < 64:
            key = key[(key >> np.uint64(shift + bits)) == (np.uint64(prefix) >> np.uint64(shift + bits))]
        if key.size:
            out[:nb] += np.bincount(((key >> np.uint64(shift)) & np.uint64(nb - 1)).astype(np.int64),
                                    minlength=nb).astype(np.uint64)
            out[nb] = min(out[nb], key.min())
            out[nb + 1] = max(out[nb + 1], key.max())
    return out


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import genodsp_amd as gd
    from oracle import cpu
    lens = [30011, 20000, 12345, 7000, 5]
    vecs = [cpu.synth_coverage(7, i, 0, n, 1 if i % 2 else 0) for i, n in enumerate(lens)]
    mine = gd.lpt_shards(lens, world)[rank]

    def allreduce(arr, op):
        # gloo has no uint64 reductions: split into two exact 32-bit halves carried in int64
        if op == "sum":
            lo = torch.from_numpy((arr & np.uint64(0xFFFFFFFF)).astype(np.int64))
            hi = torch.from_numpy((arr >> np.uint64(32)).astype(np.int64))
            dist.all_reduce(lo)
            dist.all_reduce(hi)
            return (hi.numpy().astype(np.uint64) << np.uint64(32)) + lo.numpy().astype(np.uint64)
        t = torch.from_numpy((arr ^ np.uint64(0x8000000000000000)).view(np.int64).copy())   # order-preserving
        dist.all_reduce(t, op=dist.ReduceOp.MIN if op == "min" else dist.ReduceOp.MAX)
        return t.numpy().view(np.uint64) ^ np.uint64(0x8000000000000000)

    pts = [0, 500, 50000, 99000, 100000]
    results = {}
    for window, lo, hi in ((1, -cpu.DBL_MAX, cpu.DBL_MAX), (3, 1.0, 40.0)):
        local = [vecs[i] for i in mine]
        got = gd.radix_select(lambda s, b, p: _np_histogram(local, window, lo, hi, s, b, p), pts, allreduce)
        want = cpu.percentile(vecs, pts, window, lo, hi)
        results[(window, lo, hi)] = (got[0] == want[0]) and (list(got[1]) == list(want[1]))
    q.put((rank, results, sorted(mine)))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 8])
def test_percentile_select_across_ranks(world):
    """world 8 is the contract's rank count (BASELINE configs[4]: 8 x MI355X): five chromosomes over eight ranks leave
    three ranks with nothing of their own, and they must still take part in every reduction and end with the values"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    owned = sorted(i for _, _, mine in out for i in mine)
    assert owned == [0, 1, 2, 3, 4] and len(out) == world
    for rank, results, _ in out:
        assert all(results.values()), (rank, results)
