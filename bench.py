#!/usr/bin/env python3
"""bench.py -- Gbases/s of `smooth W=101` over a 24-chromosome, 3.1 Gbp synthetic f64 signal
(BASELINE.json metric, configs[1]) on N MI355X GPUs of one node, chromosomes sharded
across ranks (LPT, no data-path collective).

    python bench.py                                  # 1 GPU
    python bench.py --gpus N                         # starts N ranks itself (torch.distributed.run), relays rank 0's line
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A step is one pass of the hot path over the whole genome: one FIR launch per chromosome
(libgenodsp_hip.so through its C ABI), inputs resident in HBM before the clock starts.
Rank 0 prints ONE JSON line.  `value` = bases of all ranks / max-over-ranks step time.
`roofline` is measured live with HIP events on the launch stream; `cpu_baseline` times the
unmodified reference (oracle/_ref, kind "reference") or, when that build is absent, the
CPU restatement (kind "port") on this host, rank 0 at N=1 only, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# hg38-like chromosome lengths fixed by SURVEY.md Appendix D (sum 3 088 269 832)
GENOME = [("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555),
          ("chr5", 181538259), ("chr6", 170805979), ("chr7", 159345973), ("chr8", 145138636),
          ("chr9", 138394717), ("chr10", 133797422), ("chr11", 135086622), ("chr12", 133275309),
          ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189), ("chr16", 90338345),
          ("chr17", 83257441), ("chr18", 80373285), ("chr19", 58617616), ("chr20", 64444167),
          ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415)]
SEED = 20240611
SAMPLE_ONE_CORE = ["chr17", "chr18", "chr19", "chr20", "chr21", "chr22", "chrY"]      # 441 Mbp: ~12 s of one core
WINDOW = 101
HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VALU_PEAK_TFLOPS = 78.6     # 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz
BYTES_PER_BASE = 16              # 8 B read + 8 B write per base for smooth
ARITHMETIC = {"hann": "hann (block sums of the window's constant and cosine parts; within one rounding per floating-point "
                      "operation of the reference, not bit-identical; opt-in --smooth=hann of the driver)",
              "fma": "fma (direct taps, one fused multiply-add each; within one rounding per operation; --smooth=fma)",
              "exact": "exact (direct taps, multiply then add: bit-identical to the reference; the driver's default)"}
WORKLOAD_KERNELS = {
    "peaks": {"fused": ["fir_fixed_extrema_kernel<101,9,FMA,true>"],
              "nofuse": ["fir_fixed_kernel<101,9,FMA>", "extrema_blocks_kernel"]},
    "morph": {"fused": ["morph_dilate_erode_kernel"], "nofuse": ["extrema_blocks_kernel", "pointwise_kernel"]},
    "percentile": {"fused": ["pc_partition_tab_kernel<2, false, true, true>", "pc_res_fixup_tab_kernel"], "nofuse": ["pc_partition_tab_kernel", "pointwise_kernel"]}}
KERNELS = {"hann": "hann_blocks_kernel<101>", "fma": "fir_fixed_kernel<101,9,true>",
           "exact": "fir_fixed_kernel<101,9,false>"}


def batch_name(kernel):
    """the one-launch-per-device form of a kernel (gdsp_*_batch): same tile code, the grid covers every vector of the rank"""
    return kernel.replace("_kernel", "_batch_kernel", 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--mode", choices=["hann", "fma", "exact"], default="hann",
                    help="arithmetic of the headline number (the other two are reported beside it): hann = block "
                         "sums of the window's constant and cosine parts, fma = direct taps with fused "
                         "multiply-add (both within one rounding per operation of the reference), exact = direct "
                         "taps, bit-identical to the reference")
    ap.add_argument("--scale", type=float, default=1.0, help="shrink every chromosome (debugging only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-workloads", action="store_true",
                    help="leave BASELINE configs[2..4] out of the line (they are timed and checked by default at one rank)")
    ap.add_argument("--sustain", type=float, default=3.0,
                    help="seconds the headline kernel is launched back to back for the `sustained` figure (0 = skip)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="debugging: every rank uses GPU 0 and the gloo backend, to exercise the N>1 code path "
                         "where only one GPU exists (numbers from such a run mean nothing)")
    ap.add_argument("--force-collectives", action="store_true",
                    help="initialise torch.distributed (nccl = RCCL) even with one rank, so that the reductions of the "
                         "percentile workload run as RCCL all-reduces on device words at world size 1 (tests)")
    ap.add_argument("--nofuse", action="store_true", help="peaks/morph workloads: one kernel per operator")
    ap.add_argument("--sharding", choices=["chromosomes", "bases"], default="chromosomes",
                    help="chromosomes = whole chromosomes dealt longest-first over the ranks (BASELINE's sharding, the "
                         "default); bases = every rank takes an equal stretch of the concatenated genome, chromosomes "
                         "cut where needed and each piece carrying the half window of neighbours it needs (smooth only)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the independent chromosomes of a rank alternate over (default 1: launches follow one "
                         "another, which is what the committed rocprofv3 per-kernel durations describe; 3 hides the drain "
                         "between kernels, +3..5 %% on the HBM-bound workloads)")
    ap.add_argument("--launch", choices=["batch", "chromosome"], default="batch",
                    help="batch = one launch per operator covers every chromosome of the rank (gdsp_*_batch, what genodsp_hip "
                         "does by default); chromosome = one launch per chromosome, one after the other (--nobatch of the driver)")
    ap.add_argument("--workload", choices=["smooth", "peaks", "morph", "percentile"], default="smooth",
                    help="smooth = BASELINE configs[1] (the metric); the others are configs[2..4], "
                         "reported in the same shape for DESIGN.md, never the driver's number")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # invoked plainly: start one fresh rank per GPU and relay rank 0's JSON line.  Nothing in THIS process has
        # touched HIP or torch (a process that has must never exec or be re-used as a rank), and it never does.
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch one rank per GPU" % (args.gpus, world))

    import torch
    import genodsp_amd as gd

    device_index = 0 if args.rehearse_on_one_gpu else local_rank
    torch.cuda.set_device(device_index)
    gd.set_device(device_index)
    dist = None
    reduce_device = "cuda"
    if world > 1 or args.force_collectives:
        import torch.distributed as dist
        if args.rehearse_on_one_gpu:
            dist.init_process_group(backend="gloo")
            reduce_device = "cpu"
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if dist is not None:
            dist.barrier()

    names = [c for c, _ in GENOME]
    lengths = [max(1, int(n * args.scale)) for _, n in GENOME]
    total_bases = sum(lengths)
    lpt_shards = gd.lpt_shards
    if args.sharding == "bases" and args.workload != "smooth":
        raise SystemExit("--sharding bases is implemented for the smooth workload")
    # a piece = (chromosome, first base held, first base owned, end of owned): whole chromosomes hold what they own;
    # a cut chromosome also holds (WINDOW-1)/2 neighbours either side of what it owns
    shards = shard_pieces(lengths, world, args.sharding, lpt_shards)
    pieces = dict(enumerate(shards[rank])) if args.sharding == "bases" else {c: (c, 0, 0, lengths[c]) for c, _, _, _ in shards[rank]}
    mine = list(pieces)
    held = {k: piece_extent(pieces[k], lengths) for k in mine}          # (lo, hi) of the bases resident for piece k

    # ---- resident signal: in/out vector per piece of this rank, generated in HBM
    stream = gd.Stream()
    vin = {k: gd.DeviceVector(held[k][1] - held[k][0]) for k in mine}
    vout = {k: gd.DeviceVector(held[k][1] - held[k][0]) for k in mine}
    for k in mine:
        gd.synth_coverage(SEED, pieces[k][0], held[k][0], held[k][1] - held[k][0], mode=1, out=vin[k], stream=stream.handle)
    stream.sync()

    # chromosomes are independent: with --streams N they alternate over N streams (all joined to the first one around
    # the timed region)
    lanes = [stream] + [gd.Stream() for _ in range(max(1, args.streams) - 1)]
    lane_of = {k: lanes[j % len(lanes)] for j, k in enumerate(mine)}

    batch = args.launch == "batch"
    items_io = gd.batch_items([vin[i] for i in mine], [vout[i] for i in mine])          # the rank's vectors as one table

    def step(mode):
        if batch:
            gd.call("gdsp_smooth_batch", items_io, len(mine), WINDOW, mode, stream.handle)
            return
        for i in mine:
            gd.smooth(vin[i], WINDOW, out=vout[i], mode=mode, stream=lane_of[i].handle)

    per_rank_ms = []                                  # filled by timed(): HIP-event ms per step of every rank, rank order

    def timed(mode, steps, warmup, step=step):
        for _ in range(warmup):
            step(mode)
        torch.cuda.synchronize()
        barrier()
        torch.cuda.synchronize()
        e0, e1 = gd.Event(), gd.Event()
        t0 = time.perf_counter()
        e0.record(stream.handle)
        for lane in lanes[1:]:
            lane.wait_event(e0)
        for _ in range(steps):
            step(mode)
        for lane in lanes[1:]:                   # the first stream ends after all of them
            done = gd.Event()
            done.record(lane.handle)
            stream.wait_event(done)
        e1.record(stream.handle)
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        wall_ms = (t1 - t0) * 1e3 / steps
        dev_ms = e0.elapsed_ms(e1) / steps          # HIP events on the launch stream
        per_rank_ms.clear()
        per_rank_ms.append(round(dev_ms, 4))
        if dist is not None:
            mine_ms = torch.tensor([dev_ms], dtype=torch.float64, device=reduce_device)
            every = [torch.zeros_like(mine_ms) for _ in range(world)]
            dist.all_gather(every, mine_ms)                                  # each rank's own device time for the step
            per_rank_ms[:] = [round(float(x[0]), 4) for x in every]
            t = torch.tensor([wall_ms, dev_ms], dtype=torch.float64, device=reduce_device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            wall_ms, dev_ms = float(t[0]), float(t[1])
        return wall_ms, dev_ms

    if args.workload != "smooth":
        other_workload(args, gd, torch, dist, rank, world, mine, lengths, total_bases, vin, vout, stream,
                       timed_fn=timed, reduce_device=reduce_device, lane_of=lane_of)
        barrier()
        if dist is not None:
            dist.destroy_process_group()
        return

    modes = {"hann": gd.FIR_HANN, "fma": gd.FIR_FMA, "exact": gd.FIR_EXACT}
    wall_ms, dev_ms = timed(modes[args.mode], args.steps, args.warmup)
    ranks_ms = list(per_rank_ms)
    others = {m: timed(modes[m], max(1, args.steps // 2), 1) for m in modes if m != args.mode}

    # ---- with more than one rank, the other split of the genome too, in the same run: equal stretches of the
    #      concatenated genome (makespan efficiency 1.000 by construction; whole chromosomes dealt longest-first reach
    #      0.965 at 8) -- one timed pass, the same arithmetic, so that one run of the driver records both
    other_split = None
    if world > 1:
        alt = "bases" if args.sharding == "chromosomes" else "chromosomes"
        alt_shards = shard_pieces(lengths, world, alt, lpt_shards)
        alt_pieces = list(alt_shards[rank])
        alt_held = [piece_extent(q, lengths) if alt == "bases" else (0, lengths[q[0]]) for q in alt_pieces]
        alt_in = [gd.DeviceVector(hi - lo) for lo, hi in alt_held]
        alt_out = [gd.DeviceVector(hi - lo) for lo, hi in alt_held]
        for q, (lo, hi), v in zip(alt_pieces, alt_held, alt_in):
            gd.synth_coverage(SEED, q[0], lo, hi - lo, mode=1, out=v, stream=stream.handle)
        stream.sync()
        alt_items = gd.batch_items(alt_in, alt_out)

        def alt_step(mode):
            if batch:
                gd.call("gdsp_smooth_batch", alt_items, len(alt_in), WINDOW, mode, stream.handle)
                return
            for a, b in zip(alt_in, alt_out):
                gd.smooth(a, WINDOW, out=b, mode=mode, stream=stream.handle)

        alt_wall, alt_dev = timed(modes[args.mode], max(1, args.steps // 2), 1, step=alt_step)
        other_split = {"sharding": alt, "value": round(total_bases / (alt_wall * 1e-3) / 1e9, 2), "unit": "Gbases/s",
                       "ms_per_step": round(alt_wall, 4), "per_rank_ms": list(per_rank_ms),
                       "bases_per_rank": [sum(b - a for _, _, a, b in sh) for sh in alt_shards]}
        del alt_in, alt_out

    # ---- parity spot check against the CPU oracle (checker only): sampled windows of the
    #      longest local chromosome, exact mode must be bit-identical, fma within tolerance
    parity = spot_check(gd, vin, vout, pieces, held, lengths, stream)

    def roofline(mode, dev_ms_per_step):
        # per launch: algorithmic bytes = 16 B/base x bases of that launch; averaged over the
        # rank with the most bases (the one that sets the step time)
        owned = [sum(b - a for _, _, a, b in sh) for sh in shards]
        busiest = max(range(world), key=lambda r: owned[r])
        bases_rank = owned[busiest]
        launches = 1 if batch else max(1, len(shards[busiest]))
        avg_launch_ms = dev_ms_per_step / launches
        achieved = BYTES_PER_BASE * bases_rank / (dev_ms_per_step * 1e-3) / 1e9
        kernel = batch_name(KERNELS[mode]) if batch else KERNELS[mode]
        r = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(achieved / HBM_PEAK_GBS, 4),
             "traffic": measured_traffic(kernel, BYTES_PER_BASE * bases_rank / launches),
             "traffic_measured": traffic_source(kernel),
             "kernel": kernel, "avg_launch_ms": round(avg_launch_ms, 4),
             "launches_per_step": launches,
             "algorithmic_bytes_per_launch": int(BYTES_PER_BASE * bases_rank / launches)}
        if mode != "hann":                           # direct evaluation: 101 multiply-adds per base on the FP64 pipe
            flops = 2.0 * WINDOW * bases_rank / (dev_ms_per_step * 1e-3) / 1e12
            r["fp64_valu_tflops"] = round(flops, 2)
            r["fp64_valu_frac"] = round(flops / FP64_VALU_PEAK_TFLOPS, 4)
        return r

    # ---- the headline kernel back to back for >= 3 s (HIP events on the launch stream): what the part sustains, beside
    #      the K-step figure the contract asks for (20 steps are 0.16 s; FP64-heavy kernels settle at a lower clock)
    sustained = None
    if args.sustain > 0:
        sus_steps = max(args.steps, int(np.ceil(args.sustain * 1e3 / max(dev_ms, 1e-3))))
        sus_wall, sus_dev = timed(modes[args.mode], sus_steps, 0)
        sustained = {"kernel": batch_name(KERNELS[args.mode]) if batch else KERNELS[args.mode], "fir_mode": args.mode,
                     "steps": sus_steps, "seconds": round(sus_wall * sus_steps * 1e-3, 3),
                     "ms_per_step": round(sus_wall, 4), "device_ms_per_step": round(sus_dev, 4),
                     "value": round(total_bases / (sus_wall * 1e-3) / 1e9, 2), "unit": "Gbases/s",
                     "frac_of_hbm_peak": round(BYTES_PER_BASE * max(sum(b - a for _, _, a, b in sh) for sh in shards)
                                               / (sus_dev * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                     "rel_diff_vs_ms_per_step": round((sus_wall - wall_ms) / wall_ms, 4),
                     "note": "launches queued back to back, one device sync at the end; rel_diff = (sustained - ms_per_step) / ms_per_step"}

    # ---- BASELINE configs[2..4] in the same process, each with its own check against the oracle (one rank: the
    #      scaling runs time the metric only)
    workloads = None
    if world == 1 and args.sharding == "chromosomes" and not args.no_workloads:
        workloads = bench_workloads(args, gd, torch, names, mine, lengths, total_bases, vin, vout, stream, timed, lane_of)

    result = {
        "metric": "Gbases/sec on smooth W=101 over 3.1 Gbp; HBM GB/s vs peak at 1/2/4/8 GPU",
        "value": round(total_bases / (wall_ms * 1e-3) / 1e9, 2),
        "unit": "Gbases/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(wall_ms, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {"workload": "smooth W=101 on 24-chrom 3.1 Gbp synthetic signal (BASELINE configs[1]); arithmetic of `value`: "
                               + ARITHMETIC[args.mode] + "; the other two arithmetics are in other_modes",
                   "window": WINDOW, "chromosomes": len(GENOME), "bases": total_bases,
                   "fir_mode": args.mode, "library": gd.lib().gdsp_version().decode(),
                   "streams": args.streams, "launch": "one launch per step covers every chromosome of the rank" if batch
                               else "one launch per chromosome",
                   "sharding": "whole chromosomes, LPT over ranks" if args.sharding == "chromosomes"
                               else "equal stretches of the concatenated genome, pieces with a half-window halo",
                   "signal": "read-depth-like x U(0.5,1.5), seed %d" % SEED},
        "roofline": roofline(args.mode, dev_ms),
        "other_modes": [{"fir_mode": m, "value": round(total_bases / (w * 1e-3) / 1e9, 2),
                         "ms_per_step": round(w, 4), "roofline": roofline(m, d)} for m, (w, d) in others.items()],
        "parity": parity,
        "per_rank_ms": ranks_ms,
        "bases_per_rank": [sum(b - a for _, _, a, b in sh) for sh in shards],
    }
    if other_split is not None:
        result["other_sharding"] = other_split
    if sustained is not None:
        result["sustained"] = sustained
    if workloads is not None:
        result["workloads"] = workloads
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(gd, lengths, names, stream)
        result["cpu_baseline_all_cores"] = cpu_baseline_all_cores(gd, lengths, names, stream)
    barrier()
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.destroy_process_group()


def self_launch(n):
    """python bench.py --gpus N outside a launcher: run `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port <free> bench.py <same arguments>` as a child and exit with its code; the
    ranks' stdout (rank 0's JSON line) and stderr are inherited."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def shard_pieces(lengths, world, sharding, lpt_shards):
    """Per rank, the pieces (chromosome, unused, first owned base, end of owned bases) it processes."""
    if sharding == "chromosomes":
        return [[(c, 0, 0, lengths[c]) for c in sh] for sh in lpt_shards(lengths, world)]
    total = sum(lengths)
    cuts = [total * r // world for r in range(world + 1)]
    shards, start = [[] for _ in range(world)], 0
    for c, n in enumerate(lengths):
        for r in range(world):
            a, b = max(cuts[r], start), min(cuts[r + 1], start + n)
            if a < b:
                shards[r].append((c, 0, a - start, b - start))
        start += n
    return shards


def piece_extent(piece, lengths):
    c, _, a, b = piece
    half = (WINDOW - 1) // 2
    return max(0, a - half), min(lengths[c], b + half)


def make_workload(gd, torch, dist, args, workload, fir, nofuse, mine, lengths, src, vout, tmp, stream, lane_of,
                  reduce_device="cuda", alone=False):
    """One of BASELINE configs[2..4] over this rank's chromosomes: -> (name, credited B/base, step(_), extra, kernels).
    src = the resident inputs (left intact), tmp = where the chain's result lands, vout = room for the unfused forms;
    fir = arithmetic of smooth in front of localmax (exact | fma); alone = the percentile without its binarize."""
    S = stream.handle
    batch = args.launch == "batch"
    n_mine = len(mine)
    io_in_out = gd.batch_items([src[i] for i in mine], [vout[i] for i in mine])
    io_in_tmp = gd.batch_items([src[i] for i in mine], [tmp[i] for i in mine])
    io_out_tmp = gd.batch_items([vout[i] for i in mine], [tmp[i] for i in mine])
    io_tmp = gd.batch_items(None, [tmp[i] for i in mine])
    # index outputs follow strict comparisons of the smoothed values: direct taps only (hann -> fma)
    mode = gd.FIR_EXACT if fir == "exact" else gd.FIR_FMA
    extra = {}
    if workload == "peaks":          # configs[2]: smooth W=101 = localmax N=11
        name, bytes_per_base = "smooth W=101 = localmax N=11", 32

        def step(_):
            if batch and nofuse:
                gd.call("gdsp_smooth_batch", io_in_out, n_mine, WINDOW, mode, S)
                gd.call("gdsp_local_extrema_batch", io_out_tmp, n_mine, 11, 1, 0.0, S)
                return
            if batch:
                gd.call("gdsp_smooth_local_extrema_batch", io_in_tmp, n_mine, WINDOW, mode, 11, 1, 0.0, S)
                return
            for i in mine:
                if nofuse:
                    gd.smooth(src[i], WINDOW, out=vout[i], mode=mode, stream=lane_of[i].handle)
                    gd.localmax(vout[i], 11, out=tmp[i], stream=lane_of[i].handle)
                else:
                    gd.smooth_local_extrema(src[i], WINDOW, 11, True, 0.0, out=tmp[i], mode=mode, stream=lane_of[i].handle)
    elif workload == "morph":        # configs[3]: dilate 1001 = erode 1001 = binarize
        name, bytes_per_base = "dilate 1001 = erode 1001 = binarize", 48
        left, right = gd.split_length(1001)

        def step(_):
            if batch and nofuse:
                gd.call("gdsp_dilate_batch", io_in_out, n_mine, left, right, 0.0, 1.0, 0.0, S)
                gd.call("gdsp_erode_batch", io_out_tmp, n_mine, left, right, 0.0, 1.0, 0.0, S)
                gd.call("gdsp_binarize_batch", io_tmp, n_mine, 0.0, 0, 1.0, 0.0, S)
                return
            if batch:
                gd.call("gdsp_dilate_erode_batch", io_in_tmp, n_mine, left, right, 0.0, 1.0, 0.0, left, right, 0.0, 1.0, 0.0,
                        1, 0.0, 0, 1.0, 0.0, S)
                return
            for i in mine:
                if nofuse:
                    gd.dilate(src[i], left, right, out=vout[i], stream=lane_of[i].handle)
                    gd.erode(vout[i], left, right, out=tmp[i], stream=lane_of[i].handle)
                    gd.binarize(tmp[i], 0.0, stream=lane_of[i].handle)
                else:
                    gd.dilate_erode(src[i], left, right, left, right, binarize=(0.0, False, 1.0, 0.0), out=tmp[i], stream=lane_of[i].handle)
    else:                                 # configs[4]: percentile 99 = binarize --threshold=percentile99
        name, bytes_per_base = ("percentile 99", 8) if alone else ("percentile 99 = binarize --threshold=percentile99", 24)
        if nofuse and not alone:
            for i in mine:
                gd.call("gdsp_memcpy_d2d", tmp[i].ptr, src[i].ptr, lengths[i] * 8, gd._sp(S))
        device_allreduce = None
        if dist is not None:
            # the path's only collective, on the words where the library left them in HBM: the device address is
            # wrapped as a torch tensor (no copy) and all-reduced by RCCL on the library's own stream
            class _Words:
                def __init__(self, ptr, count):
                    self.__cuda_array_interface__ = {"shape": (count,), "typestr": "<i8", "data": (int(ptr), False), "version": 2}

            TOP = -(1 << 63)                 # u64 order as i64 order: flip the top bit around a min / max

            def device_allreduce(ptr, count, op, stream_handle):
                ext = torch.cuda.ExternalStream(int(stream_handle)) if stream_handle else torch.cuda.current_stream()
                with torch.cuda.stream(ext):
                    t = torch.as_tensor(_Words(ptr, count), device="cuda")
                    if op != "sum":
                        t.bitwise_xor_(TOP)
                    how = {"sum": dist.ReduceOp.SUM, "min": dist.ReduceOp.MIN, "max": dist.ReduceOp.MAX}[op]
                    if reduce_device == "cpu":           # one-GPU rehearsal over gloo: the only case with a host hop
                        h = t.cpu()
                        dist.all_reduce(h, op=how)
                        t.copy_(h)
                    else:
                        dist.all_reduce(t, op=how)
                    if op != "sum":
                        t.bitwise_xor_(TOP)

        def step(_):
            if not nofuse and not alone:
                # both operators in the percentile's own read of the signal (gdsp_percentiles_binarize)
                cnt, vals, _outs, one_pass = gd.percentile_binarize([src[i] for i in mine], [99000], which=0, outs=[tmp[i] for i in mine],
                                                                    device_allreduce=device_allreduce, stream=S)
                extra["percentile99"], extra["sampled"], extra["binarize_in_one_pass"] = vals[0], cnt, bool(one_pass)
                st = gd.percentile_stats()
                extra["percentile_route"] = {gd.SELECT_RADIX: "radix", gd.SELECT_BRACKET: "bracket"}.get(st["route"])
                extra["percentile_stats"] = st
                return
            cnt, vals = gd.percentile([src[i] for i in mine], [99000], device_allreduce=device_allreduce, stream=S)
            extra["percentile99"], extra["sampled"] = vals[0], cnt
            st = gd.percentile_stats()
            extra["percentile_route"] = {gd.SELECT_RADIX: "radix", gd.SELECT_BRACKET: "bracket"}.get(st["route"])
            extra["percentile_stats"] = st
            if alone:
                return
            if batch:
                gd.call("gdsp_binarize_batch", io_tmp, n_mine, float(vals[0]), 0, 1.0, 0.0, S)
                return
            for i in mine:
                gd.binarize(tmp[i], vals[0], stream=lane_of[i].handle)
    kernels = (["pc_partition_tab_kernel<2, false, true, false>"] if alone
               else WORKLOAD_KERNELS[workload]["nofuse" if nofuse else "fused"])
    if workload == "peaks":
        kernels = [k.replace("FMA", "false" if fir == "exact" else "true") for k in kernels]
        filtered = os.environ.get("GDSP_PEAKS_FILTER") not in (("0",) if fir == "exact" else ("0", "exact"))
        if not nofuse and filtered:
            # the filtered route (gdsp_peaks.hip): block sums + interval test, then exact taps for what stays undecided
            fma = "false" if fir == "exact" else "true"
            kernels = ["peaks_filter_kernel<101, %s, true, 5, false>" % fma, "peaks_exact_kernel<101, %s, true>" % fma,
                       "peaks_probe_kernel<101, true, 5, true>", "fir_fixed_extrema_gated_kernel (leaves at once)"]
    if batch:
        kernels = [k if k.startswith(("pc_", "peaks_", "fir_fixed_extrema_gated")) else batch_name(k) for k in kernels]
    return name, bytes_per_base, step, extra, kernels


def workload_roofline(gd, args, world, mine, lengths, dev_ms, bytes_per_base, moved_per_base, kernels):
    """achieved / frac count the bytes that cross HBM; credited_by_survey_8d the 16 B per operator executed"""
    batch = args.launch == "batch"
    bases_rank = max(sum(lengths[i] for i in sh) for sh in gd.lpt_shards(lengths, world))
    achieved = moved_per_base * bases_rank / (dev_ms * 1e-3) / 1e9            # bytes that actually cross HBM
    credited = bytes_per_base * bases_rank / (dev_ms * 1e-3) / 1e9            # 16 B per operator executed
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": measured_traffic(kernels[0], moved_per_base * bases_rank / (1 if batch else max(1, len(mine)))),
            "traffic_measured": traffic_source(kernels[0]),
            "kernel": kernels[0], "kernels": kernels,
            "hbm_bytes_per_base_moved": moved_per_base,
            "credited_by_survey_8d": {"bytes_per_base": bytes_per_base, "achieved": round(credited, 1),
                                      "frac": round(credited / HBM_PEAK_GBS, 4)},
            "note": "achieved / frac count the bytes that cross HBM (a fused chain moves 16 B/base in all); "
                    "credited_by_survey_8d counts 16 B per operator executed (8 B for the percentile pass), "
                    "fused or not, and may therefore exceed what HBM delivers"}


def other_workload(args, gd, torch, dist, rank, world, mine, lengths, total_bases, vin, vout, stream, timed_fn, lane_of,
                   reduce_device="cuda"):
    """`--workload peaks|morph|percentile`: ONE of BASELINE configs[2..4] as the whole line (tools, DESIGN.md, the
    multi-rank tests); the driver's default run carries all of them under `workloads` (bench_workloads)."""
    tmp = {i: gd.DeviceVector(lengths[i]) for i in mine}
    fir = "exact" if args.mode == "exact" else "fma"
    name, bytes_per_base, step, extra, kernels = make_workload(gd, torch, dist, args, args.workload, fir, args.nofuse, mine, lengths,
                                                               vin, vout, tmp, stream, lane_of, reduce_device)
    wall_ms, dev_ms = timed_fn(None, args.steps, args.warmup, step=step)
    moved_per_base = 16 if not args.nofuse else bytes_per_base        # a fused chain moves 8 B in and 8 B out per base in all
    batch = args.launch == "batch"
    result = {"metric": "Gbases/sec on %s over 3.1 Gbp" % name, "value": round(total_bases / (wall_ms * 1e-3) / 1e9, 2),
              "unit": "Gbases/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": round(wall_ms, 4), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
              "dtype": "f64", "data": "synthetic",
              "config": {"workload": name + " on 24-chrom 3.1 Gbp synthetic signal", "bases": total_bases,
                         "fir_mode": fir, "fused": not args.nofuse,
                         "library": gd.lib().gdsp_version().decode(),
                         "collectives": (None if dist is None else "gloo, host copy (one-GPU rehearsal)" if reduce_device == "cpu"
                                         else "rccl (torch.distributed nccl backend), device words"),
                         "streams": args.streams, "launch": "one launch per operator covers every chromosome of the rank" if batch
                                    else "one launch per operator and chromosome",
                         "sharding": "whole chromosomes, LPT over ranks"},
              "roofline": workload_roofline(gd, args, world, mine, lengths, dev_ms, bytes_per_base, moved_per_base, kernels)}
    result.update(extra)
    if rank == 0:
        print(json.dumps(result))


def bench_workloads(args, gd, torch, names, mine, lengths, total_bases, vin, vout, stream, timed_fn, lane_of):
    """BASELINE configs[2..4] in the driver's own run (one rank): each timed like the headline (same harness, same
    resident signal, the reference's arithmetic) and checked against the CPU oracle right here.
      configs[2]  smooth W=101 = localmax N=11, exact, fused (the filtered route) -- on the real-valued genome and on
                  read depth (piecewise-constant integers, what coverage pipelines carry)
      configs[3]  dilate 1001 = erode 1001 = binarize, fused
      configs[4]  percentile 99 = binarize --threshold=percentile99, fused; and percentile 99 alone
    A step = the chain over the 24 chromosomes (genodsp.c:900-936: every operator over every chromosome)."""
    from oracle import cpu                           # the checker; never on the measured path
    steps, warmup = max(3, min(args.steps, 10)), 2
    tmp = {i: gd.DeviceVector(lengths[i]) for i in mine}
    depth = {i: gd.DeviceVector(lengths[i]) for i in mine}
    for i in mine:
        gd.synth_coverage(SEED, i, 0, lengths[i], mode=0, out=depth[i], stream=stream.handle)
    stream.sync()
    k = max(mine, key=lambda i: lengths[i])          # the chromosome the window checks look at
    n = lengths[k]
    rng = np.random.default_rng(2)
    span = min(4096, n)
    starts = sorted(set([0, n - span] + [int(s) for s in rng.integers(0, max(1, n - span), 6)]))
    left, right = gd.split_length(1001)

    def windows_equal(out_vec, synth_mode, margin, chain):
        """chain(oracle, stretch) on [s-margin, s+m+margin) of the regenerated input, cut to [s, s+m), against what the
        HIP path left in out_vec: every bit"""
        same = True
        for s in starts:
            m = min(span, n - s)
            xlo, xhi = max(0, s - margin), min(n, s + m + margin)
            want = chain(cpu.synth_coverage(SEED, k, xlo, xhi - xlo, synth_mode))[s - xlo:s - xlo + m]
            got = out_vec.buf.download(np.float64, m, out_vec.offset + 8 * s)
            same = same and got.tobytes() == np.ascontiguousarray(want, np.float64).tobytes()
        return {"chromosome": names[k], "windows": len(starts), "window_len": span, "bit_identical": bool(same), "ok": bool(same),
                "against": "oracle (CPU restatement of the reference's loops) on the regenerated stretches"}

    peaks_chain = lambda x: cpu.local_extrema(cpu.smooth(x, WINDOW), 11, True, 0.0)
    morph_chain = lambda x: cpu.binarize(cpu.erode(cpu.dilate(x, left, right), left, right), 0.0)
    plan = [("peaks", "exact", vin, 1, "real-valued genome (the bench signal)", lambda: windows_equal(tmp[k], 1, 64, peaks_chain)),
            ("peaks", "exact", depth, 0, "read depth (piecewise-constant integers)", lambda: windows_equal(tmp[k], 0, 64, peaks_chain)),
            ("morph", None, vin, 1, "real-valued genome (the bench signal)", lambda: windows_equal(tmp[k], 1, 2304, morph_chain)),
            ("percentile", None, vin, 1, "real-valued genome (the bench signal)", None),
            ("percentile_alone", None, vin, 1, "real-valued genome (the bench signal)", None)]
    out = []
    counts = {}                                      # (value) -> (below, equal) over the genome, counted on the host once
    for workload, fir, src, synth_mode, signal, check in plan:
        alone = workload == "percentile_alone"
        wl = "percentile" if alone else workload
        name, bytes_per_base, step, extra, kernels = make_workload(gd, torch, None, args, wl, fir, False, mine, lengths, src, vout, tmp,
                                                                   stream, lane_of, alone=alone)
        wall_ms, dev_ms = timed_fn(None, steps, warmup, step=step)
        moved = 8 if alone else 16
        entry = {"workload": name + " on the 24-chrom 3.1 Gbp " + signal + (", smooth in the reference's arithmetic (exact), "
                             "fused" if workload == "peaks" else ", one read of the signal" if alone else ", fused"),
                 "config": "BASELINE configs[%d]" % {"peaks": 2, "morph": 3}.get(workload, 4),
                 "value": round(total_bases / (wall_ms * 1e-3) / 1e9, 2), "unit": "Gbases/s",
                 "ms_per_step": round(wall_ms, 4), "device_ms_per_step": round(dev_ms, 4), "steps": steps, "warmup": warmup,
                 "roofline": workload_roofline(gd, args, 1, mine, lengths, dev_ms, bytes_per_base, moved, kernels)}
        if check is not None:
            entry["parity"] = check()
        else:
            entry["parity"] = percentile_check(gd, cpu, names, mine, lengths, vin, tmp, k, starts, span, extra, counts,
                                               binarized=not alone)
        for key in ("percentile99", "sampled", "percentile_route", "binarize_in_one_pass"):
            if key in extra:
                entry[key] = extra[key]
        out.append(entry)
    return out


def percentile_check(gd, cpu, names, mine, lengths, vin, tmp, k, starts, span, extra, counts, binarized):
    """The HIP path's genome-wide percentile held to the reference's definition without sorting 3.1 G values on the
    host: the value is the order statistic of rank r = floor(count * p / 100) (percentile.c:587-589 as restated in
    oracle/gdsp_oracle.c:orc_percentile, r clamped to count-1) exactly when  #(x < V) <= r < #(x <= V)  -- counted on the
    host over every base of every chromosome, copied back from HBM.  The fused binarize output is then held to the
    oracle's binarize at that value on sampled windows."""
    from concurrent.futures import ThreadPoolExecutor
    V = extra["percentile99"]
    if V not in counts:
        def count(x):
            return int(np.count_nonzero(x < V)), int(np.count_nonzero(x == V))
        below = equal = 0
        with ThreadPoolExecutor(max(1, min(8, len(os.sched_getaffinity(0))))) as pool:
            pending = []
            for i in mine:                           # copies back one after the other, counting in the pool behind them
                pending.append(pool.submit(count, vin[i].numpy()))
                while len(pending) > 4:
                    b, e = pending.pop(0).result()
                    below, equal = below + b, equal + e
            for f in pending:
                b, e = f.result()
                below, equal = below + b, equal + e
        counts[V] = (below, equal)
    below, equal = counts[V]
    population = sum(lengths[i] for i in mine)
    r = min(population - 1, int(float(population * 99000) / (100.0 * 1000)))
    ok = bool(extra["sampled"] == population and below <= r < below + equal)
    res = {"percentile99": V, "population": population, "rank": r, "values_below": below, "values_equal": equal,
           "is_the_order_statistic": ok,
           "against": "the reference's rank (percentile.c:587-589) by counting every base on the host"}
    if binarized:
        n = lengths[k]
        same = True
        for s in starts:
            m = min(span, n - s)
            want = cpu.binarize(cpu.synth_coverage(SEED, k, s, m, 1), V)
            got = tmp[k].buf.download(np.float64, m, tmp[k].offset + 8 * s)
            same = same and got.tobytes() == np.ascontiguousarray(want, np.float64).tobytes()
        res.update({"binarize_chromosome": names[k], "binarize_windows": len(starts), "binarize_bit_identical": bool(same)})
        ok = ok and same
    res["ok"] = bool(ok)
    return res


def measured_traffic(kernel, algorithmic_bytes_per_launch):
    """HBM bytes per launch: the FETCH_SIZE/WRITE_SIZE ratio to algorithmic bytes measured for this
    kernel with rocprofv3 --pmc (profiles/traffic.json, gfx950 corrections applied there) scaled to
    this run's average launch; null when no such measurement is committed."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    k = _traffic_entry(path, kernel)
    ratio = None if k is None else k.get("hbm_bytes_over_algorithmic")
    return None if ratio is None else int(ratio * algorithmic_bytes_per_launch)


def _traffic_entry(path, kernel):
    """the entry of profiles/traffic.json for this kernel: same name up to blanks, else the first one it is a prefix of"""
    with open(path) as f:
        kernels = json.load(f).get("kernels", {})
    squeeze = lambda x: x.replace(" ", "")
    for name, entry in kernels.items():
        if squeeze(name) == squeeze(kernel):
            return entry
    for name, entry in kernels.items():
        if squeeze(name).startswith(squeeze(kernel)):
            return entry
    return None


def traffic_source(kernel):
    """which rocprofv3 PMC run (and library build) the traffic ratio of `kernel` comes from: a stale ratio shows here"""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(path):
        return None
    k = _traffic_entry(path, kernel)
    return None if k is None else {"source": k.get("source"), "library": k.get("library")}


def spot_check(gd, vin, vout, pieces, held, lengths, stream):
    from oracle import cpu                           # the checker; never on the measured path
    k = max(pieces, key=lambda q: pieces[q][3] - pieces[q][2])
    c, _, a, b = pieces[k]                           # owned bases [a,b) of chromosome c, resident [lo,hi)
    lo, hi = held[k]
    n = lengths[c]
    rng = np.random.default_rng(1)
    half = (WINDOW - 1) // 2
    span = min(4096, b - a)
    starts = [a, b - span] + [int(s) for s in rng.integers(a, max(a + 1, b - span), 6)]
    taps = cpu.hann_window(WINDOW)
    out = {"chromosome": GENOME[c][0], "owned": [a, b], "windows": len(starts), "window_len": span}
    worst = {"fma": 0.0, "hann": 0.0}
    exact_ok = True
    for mode, key in ((gd.FIR_EXACT, "exact"), (gd.FIR_FMA, "fma"), (gd.FIR_HANN, "hann")):
        gd.smooth(vin[k], WINDOW, out=vout[k], mode=mode, stream=stream.handle)
        stream.sync()
        for s in starts:
            m = min(span, b - s)
            xlo, xhi = max(0, s - half), min(n, s + m + half)
            x = cpu.synth_coverage(SEED, c, xlo, xhi - xlo, 1)
            xp = np.concatenate([np.zeros(half - (s - xlo)), x, np.zeros(half - (xhi - (s + m)))])
            # zero-padded FIR of the padded stretch, then cut the m outputs that belong to [s, s+m)
            want = cpu.fir(xp, taps)[half:half + m] if xp.size > 2 * half else cpu.fir(xp, taps)
            got = vout[k].buf.download(np.float64, m, vout[k].offset + 8 * (s - lo))
            if key == "exact":
                exact_ok = exact_ok and (got.tobytes() == want.tobytes())
            else:
                scale = cpu.fir(np.abs(xp), taps)[half:half + m]
                bound = WINDOW * 2.0 ** -52 * scale
                worst[key] = max(worst[key], float(np.max(np.abs(got - want) / np.maximum(bound, 1e-300))))
    out["exact_bit_identical"] = bool(exact_ok)
    out["bound"] = "W * 2^-52 * sum|w_k v_k| per output (one rounding per floating-point operation)"
    out["fma_worst_err_over_bound"] = round(worst["fma"], 4)
    out["hann_worst_err_over_bound"] = round(worst["hann"], 4)
    out["ok"] = bool(exact_ok and worst["fma"] <= 1.0 and worst["hann"] <= 1.0)
    return out


def cpu_baseline(gd, lengths, names, stream):
    """Time the CPU path on a bounded sample: smooth W=101 over chr17..chr22 + chrY (441 Mbp at full scale,
    about 12 s of one core), one thread, the same synthetic signal copied back from HBM."""
    from oracle import cpu, ref
    sample = [names.index(c) for c in SAMPLE_ONE_CORE]
    vecs = {}
    for i in sample:
        d = gd.synth_coverage(SEED, i, 0, lengths[i], mode=1, stream=stream.handle)
        stream.sync()
        vecs[i] = d.numpy()
    bases = sum(lengths[i] for i in sample)
    if ref.available():
        g = ref.Genome([(names[i], lengths[i]) for i in sample])
        for i in sample:
            g.set(names[i], vecs[i])
        t0 = time.perf_counter()
        g.run("= smooth W=%d" % WINDOW)
        dt = time.perf_counter() - t0
        outs = {i: g.get(names[i]) for i in sample}
        g.close()
        kind = "reference"
    else:
        t0 = time.perf_counter()
        outs = {i: cpu.smooth(vecs[i], WINDOW) for i in sample}
        dt = time.perf_counter() - t0
        kind = "port"
    # the same chromosomes through the HIP path in exact mode must match every bit
    same = True
    for i in sample:
        d = gd.DeviceVector.from_numpy(vecs[i])
        got = gd.smooth(d, WINDOW, mode=gd.FIR_EXACT, stream=stream.handle).numpy()
        same = same and (got.tobytes() == outs[i].tobytes())
    return {"value": round(bases / dt / 1e9, 5), "unit": "Gbases/s", "cores": 1, "kind": kind,
            "sample": "smooth W=101 on %s..%s+%s (%d bases) of the same synthetic signal, %.1f s"
                      % (SAMPLE_ONE_CORE[0], SAMPLE_ONE_CORE[-2], SAMPLE_ONE_CORE[-1], bases, dt),
            "hip_exact_bit_identical_on_sample": bool(same)}


def cpu_quota_cores():
    """the CPU share of this container in cores (cgroup v2 cpu.max or v1 cfs quota), None when unlimited / unknown"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            return None if q == "max" else max(1, int(int(q) / int(per)))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
            q, per = int(f.read()), int(g.read())
            return None if q <= 0 else max(1, q // per)
    except Exception:
        return None


def cpu_baseline_all_cores(gd, lengths, names, stream):
    """the CPU restatement (oracle port) over every host core this process may use: pthreads inside
    liboracle.so, thread t of T taking the t-th stretch of every chromosome of the sample (neighbours read in place),
    outputs allocated and touched before the clock starts (page faults of 6.8 GB of fresh output are not smoothing).
    Reported beside the single-threaded reference, never instead of it."""
    from oracle import cpu
    nproc = os.cpu_count() or 1
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else nproc     # what `nproc` prints
    quota = cpu_quota_cores()
    if quota is not None:
        cores = max(1, min(cores, quota))          # a container's CPU share: more threads than that only take turns
    if os.environ.get("GDSP_HOST_CORES"):
        cores = min(cores, int(os.environ["GDSP_HOST_CORES"]))
    sample = [names.index(c) for c in ("chr13", "chr14", "chr15", "chr16", "chr17", "chr18", "chr19", "chr20", "chr21", "chr22", "chrY")]
    vecs = []
    for i in sample:
        d = gd.synth_coverage(SEED, i, 0, lengths[i], mode=1, stream=stream.handle)
        stream.sync()
        vecs.append(d.numpy())
    outs = [np.zeros_like(v) for v in vecs]                                # allocated and touched
    bases = sum(lengths[i] for i in sample)
    t0 = time.perf_counter()
    outs, started = cpu.smooth_threads(vecs, WINDOW, cores, outs)
    dt = time.perf_counter() - t0
    # ... and they are the single-threaded loop's bits (a stretch of the last chromosome)
    same = outs[-1][:200000].tobytes() == cpu.smooth(vecs[-1][:200050], WINDOW)[:200000].tobytes()
    return {"value": round(bases / dt / 1e9, 5), "unit": "Gbases/s", "cores": started, "kind": "port",
            "host_logical_cpus": nproc, "usable_by_this_process": cores, "cgroup_cpu_quota_cores": quota,
            "same_bits_as_one_thread": bool(same),
            "sample": "smooth W=101 on chr13..chr22+chrY (%d bases), %d pthreads each taking one stretch of every chromosome (every "
                      "core this process may use; kind \"port\" = the CPU restatement, the reference itself is single-threaded), "
                      "outputs preallocated, %.2f s" % (bases, started, dt)}



if __name__ == "__main__":
    main()
