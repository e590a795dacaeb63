"""GPU: the one-process-per-GPU path of bench.py with two ranks sharing the one visible GPU (gloo for the
reductions).  What is checked is the N>1 plumbing -- LPT sharding of the chromosomes, the reduction
callback of gdsp_percentiles (counts, histograms and flags summed over ranks so that every rank takes
the same decisions), max-over-ranks timing -- not a speed."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _bench(ranks, workload, scale, extra=()):
    common = ["bench.py", "--gpus", str(ranks), "--workload", workload, "--scale", str(scale), "--steps", "1",
              "--warmup", "0", "--no-cpu-baseline", "--sustain", "0.2"] + list(extra)
    if ranks == 1:
        cmd = [sys.executable] + common
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks),
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port())] + common + ["--rehearse-on-one-gpu"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                      # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def test_plain_invocation_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (what the driver's scaling run may do): bench.py starts
    the two ranks itself, before anything touches HIP, and relays rank 0's one JSON line."""
    cmd = [sys.executable, "bench.py", "--gpus", "2", "--workload", "percentile", "--scale", "0.08", "--steps", "1",
           "--warmup", "0", "--no-cpu-baseline", "--rehearse-on-one-gpu"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    two = json.loads(lines[0])
    one = _bench(1, "percentile", 0.08)
    assert two["n_gpus"] == 2 and two["percentile99"] == one["percentile99"] and two["sampled"] == one["sampled"]


def test_device_reduction_hook_through_rccl_at_world_size_one():
    """One rank, backend nccl (= RCCL): the percentile workload's reductions run as torch.distributed all-reduces on
    the library's device words (no host copy) -- the code path of the N-GPU run, exercised on the one-GPU box."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), "bench.py", "--gpus", "1", "--workload", "percentile", "--scale", "0.08",
           "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--force-collectives"]
    p = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    got = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][0])
    one = _bench(1, "percentile", 0.08)
    assert got["percentile99"] == one["percentile99"] and got["sampled"] == one["sampled"]
    assert got["config"]["collectives"] == "rccl (torch.distributed nccl backend), device words"
    assert got["percentile_stats"]["resident"] == 1 and got["binarize_in_one_pass"] is True, got["percentile_stats"]


def test_percentile_over_two_ranks_equals_one_rank():
    # 0.08 of the genome = 247 Mbp: above the 2^24 values where the bracketing route starts
    one = _bench(1, "percentile", 0.08)
    two = _bench(2, "percentile", 0.08)
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["sampled"] == one["sampled"] == one["config"]["bases"]
    assert two["percentile99"] == one["percentile99"]
    assert two["percentile_route"] == one["percentile_route"] == "bracket"
    # both are decided on the device with one read-back: across ranks every digit pass and the counting pass are cut at
    # their reduction (count, all-reduce of the device words through the hook, pick) -- gdsp_percentile.hip: pc_resident
    for r in (one, two):
        assert r["percentile_stats"]["resident"] == 1 and r["percentile_stats"]["fallbacks"] == 0, r["percentile_stats"]
        assert r["binarize_in_one_pass"] is True
    assert two["percentile_stats"]["population"] == one["percentile_stats"]["population"]
    assert two["percentile_stats"]["sample"] == one["percentile_stats"]["sample"]


def test_percentile_alone_over_two_ranks_equals_one_rank():
    """the same without the fused binarize (gdsp_percentiles), two ranks sharing the GPU against one"""
    one = _bench(1, "percentile", 0.08, ["--nofuse"])
    two = _bench(2, "percentile", 0.08, ["--nofuse"])
    assert two["percentile99"] == one["percentile99"] and two["sampled"] == one["sampled"]
    for r in (one, two):
        assert r["percentile_stats"]["resident"] == 1 and r["percentile_stats"]["fallbacks"] == 0, r["percentile_stats"]


def test_smooth_over_two_ranks_reports_the_whole_job():
    two = _bench(2, "smooth", 0.02)
    assert two["n_gpus"] == 2 and two["parity"]["ok"] and two["scaling"] == "strong"
    assert two["config"]["bases"] == sum(max(1, int(n * 0.02)) for n in
                                         [248956422, 242193529, 198295559, 190214555, 181538259, 170805979, 159345973,
                                          145138636, 138394717, 133797422, 135086622, 133275309, 114364328, 107043718,
                                          101991189, 90338345, 83257441, 80373285, 58617616, 64444167, 46709983,
                                          50818468, 156040895, 57227415])


def test_smooth_over_equal_stretches_of_the_genome():
    """--sharding bases: chromosomes are cut so that every rank owns the same number of bases; a cut piece holds
    the half window of neighbours it needs, and its owned outputs must be those of the whole chromosome."""
    one = _bench(1, "smooth", 0.02, ["--sharding", "bases"])
    two = _bench(2, "smooth", 0.02, ["--sharding", "bases"])
    for r in (one, two):
        assert r["parity"]["ok"] and r["parity"]["exact_bit_identical"]
        assert "equal stretches" in r["config"]["sharding"]
    assert two["roofline"]["algorithmic_bytes_per_launch"] * two["roofline"]["launches_per_step"] <= 16 * (two["config"]["bases"] // 2 + 1)


def test_smooth_over_two_ranks_reports_both_splits_and_every_ranks_time():
    """At N > 1 one run records both ways of splitting the genome -- whole chromosomes dealt longest-first (the metric)
    and equal stretches of bases -- with each rank's own HIP-event milliseconds, so that a single run of the scaling
    bench shows the load balance of either."""
    two = _bench(2, "smooth", 0.02)
    assert two["n_gpus"] == 2 and len(two["per_rank_ms"]) == 2 and all(t > 0 for t in two["per_rank_ms"])
    assert sum(two["bases_per_rank"]) == two["config"]["bases"]
    alt = two["other_sharding"]
    assert alt["sharding"] == "bases" and len(alt["per_rank_ms"]) == 2 and alt["value"] > 0
    assert sum(alt["bases_per_rank"]) == two["config"]["bases"]
    assert max(alt["bases_per_rank"]) - min(alt["bases_per_rank"]) <= 1            # equal shares
    assert two["parity"]["ok"]


def test_default_line_carries_the_other_baseline_configs_and_a_sustained_figure():
    """The driver's one-rank run also times BASELINE configs[2..4] (`workloads`, each held to the oracle in the run) and
    the headline kernel back to back (`sustained`); shrunk here, the keys and the checks are the full run's."""
    one = _bench(1, "smooth", 0.02)
    assert one["sustained"]["steps"] >= 1 and one["sustained"]["ms_per_step"] > 0
    assert abs(one["sustained"]["rel_diff_vs_ms_per_step"]) < 10
    got = one["workloads"]
    assert [w["config"] for w in got] == ["BASELINE configs[2]", "BASELINE configs[2]", "BASELINE configs[3]",
                                          "BASELINE configs[4]", "BASELINE configs[4]"]
    for w in got:
        assert w["parity"]["ok"] is True, w
        assert w["value"] > 0 and w["roofline"]["frac"] > 0 and "credited_by_survey_8d" in w["roofline"]
    assert got[3]["percentile99"] == got[4]["percentile99"] and got[3]["parity"]["is_the_order_statistic"]
    two = _bench(2, "smooth", 0.02)
    assert "workloads" not in two and "sustained" in two           # the scaling runs time the metric only


def test_three_ranks_share_the_gpu():
    """More ranks than two beside this process, within what the test box lets share its one GPU (its guard allows six GPU
    processes in all, and five ranks beside the test runner tripped it; the contract's eight run on CPU in
    tests/test_distributed_cpu.py and as eight device shards of the C driver in tests/test_cli_genome.py /
    test_cli_seams.py): the metric with both splits over an odd number of ranks, and percentile's reductions over three
    ranks arriving at the one-rank value through the resident route."""
    three = _bench(3, "smooth", 0.01)
    assert three["n_gpus"] == 3 and len(three["per_rank_ms"]) == 3 and len(three["bases_per_rank"]) == 3
    assert sum(three["bases_per_rank"]) == three["config"]["bases"] and three["parity"]["ok"]
    alt = three["other_sharding"]
    assert alt["sharding"] == "bases" and len(alt["per_rank_ms"]) == 3
    assert max(alt["bases_per_rank"]) - min(alt["bases_per_rank"]) <= 1
    one = _bench(1, "percentile", 0.08)
    pc = _bench(3, "percentile", 0.08)
    assert pc["n_gpus"] == 3 and pc["percentile99"] == one["percentile99"] and pc["sampled"] == one["sampled"]
    assert pc["percentile_stats"]["resident"] == 1 and pc["percentile_stats"]["fallbacks"] == 0, pc["percentile_stats"]
