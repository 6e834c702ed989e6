#!/usr/bin/env python3
"""bench.py — the reference's headline workload on MI355X.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N ...        (starts its own N ranks as child processes, see launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One step = one pass of the whole hot path (outer seed stream, per-pair planning,
offset scan, emit kernel, metadata, counters) over one batch: BASELINE.json
configs[1] — minimal-short 150 bp PE on a 100 Mbp synthetic genome, 100 M reads
per GPU, inputs (the 2-bit packed reference) resident in HBM.  With N GPUs the
job is N x 100 M reads of ONE run (same seed): rank r simulates pair-index range
[r*50M, (r+1)*50M) — weak scaling, no data-path collective; one all-reduce of
the 8 run counters at the end of every step (RCCL over xGMI).

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel
(k_emit_philox in the default counter mode): algorithmic bytes per launch / its
HIP-event duration measured on the engine's stream.  `cpu_baseline` is the CPU oracle (a port of the
reference algorithm; the Rust reference cannot be built here) timed on this
host on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.29 TB/s measured copy)


def launch_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes of
    `python -m torch.distributed.run` (one rank per GPU, rendezvous on 127.0.0.1 at a free port), pass their
    output through (rank 0 prints the JSON line) and return the launcher's exit code.  Called before this
    process imports torch, so nothing here has initialised a GPU; nothing is exec'ed."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("OMP_NUM_THREADS", "1")             # (the launcher would set it anyway; cpu_baseline sets its own)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + list(argv)
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--reads", type=int, default=100_000_000, help="reads per GPU per step")
    ap.add_argument("--genome-bases", type=int, default=100_000_000)
    ap.add_argument("--profile", default="minimal-short", choices=["minimal-short", "perfect-short", "minimal-long", "custom-short", "custom-long"],
                    help="custom-long: a synthetic simmrd-shaped long-read model (k = 7, every 7-mer listed, 1000 modelled "
                         "positions) through the long-read path with per-read lengths (BASELINE config 5's profile)")
    ap.add_argument("--length-normal", default="20000,4000", help="custom-long: read_length_mean,read_length_std of the model")
    ap.add_argument("--gamma", default="8000,6000", help="minimal-long: gamma mean,std of the read length (BASELINE config 3)")
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--rng", default="philox", choices=["reference", "philox"],
                    help="philox: Philox4x32-10 counter mode for the per-base draws (north_star's design, tolerance "
                         "parity); reference: the reference's own ChaCha12 streams (bit-exact, slower)")
    ap.add_argument("--plan", default="philox", choices=["reference", "philox"],
                    help="with --rng philox, where the PLAN's draws (contig, seeds, lengths, positions) come from: philox = Philox "
                         "counters too (SIMMR_RNG_PHILOX_FULL: minimal-short and minimal-long; the default), reference = the "
                         "reference's ChaCha12 streams (SIMMR_RNG_PHILOX: same positions and lengths as the reference run)")
    ap.add_argument("--plan-overlap", action="store_true",
                    help="the engine plans on a stream of its own into a second set of plan buffers (simmr_engine_set_plan_overlap): "
                         "the plan of step k + 1 then runs beside the emit of step k.  Off by default — the emit kernel's time, "
                         "which `roofline` is about, is then that of the kernel alone; the default line times ten steps with it "
                         "as `with_plan_overlap`")
    ap.add_argument("--no-overlap-side", action="store_true",
                    help="skip the `with_plan_overlap` side measurement (kernel-trace runs: its launches share the device and "
                         "would be averaged into the kernels' own times)")
    ap.add_argument("--no-other-mode", action="store_true", help="skip the untimed side measurements (the other rng mode, the FASTQ text)")
    ap.add_argument("--through-fastq", action="store_true",
                    help="one step = plan + FASTQ sizing + emit straight into FASTQ text resident in HBM (simmr_fastq_plan_direct / "
                         "simmr_emit_fastq: what the reference's run produces, main.rs:180-206), instead of the SoA columns")
    ap.add_argument("--layout", default=None, choices=["compact", "slot16"],
                    help="slot16 (the default where it is offered: counter mode, minimal profiles): every read in a 16-byte-aligned "
                         "slot of seq / qual (SIMMR_SLOT16, include/simmr_hip.h), the emit kernel then writes whole aligned 16-byte "
                         "groups only; compact: byte streams without gaps (the other one is timed once as `other_layout`)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-reads", type=int, default=None,
                    help="cpu_baseline sample of the paired-end profiles (default 20 M reads, 400 000 for custom-short)")
    ap.add_argument("--cpu-sample-long-reads", type=int, default=None,
                    help="cpu_baseline sample of the long-read profiles (BASELINE.md B4; default 400 000 reads, 1000 for custom-long)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo only for rehearsing N>1 on a single GPU (all ranks then share --rehearse-device)")
    ap.add_argument("--rehearse-device", type=int, default=None)
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # started plainly with --gpus N: this process has touched no device yet and never will
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))

    import numpy as np
    import torch
    import torch.distributed as dist

    from simmr_amd import MinimalLongErrorProfile, MinimalShortErrorProfile, PerfectShortErrorProfile, _abi
    from simmr_amd.engine import Engine, Reads

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {args.gpus}` (it launches "
                         f"its own ranks) or under torch.distributed.run with --nproc-per-node {args.gpus}")
    if args.rehearse_device is not None:
        local_rank = args.rehearse_device
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    eng = Engine(local_rank)
    eng.stage_synthetic(0, [args.genome_bases], 2)  # SplitMix64(seed=2)
    long_mode = args.profile in ("minimal-long", "custom-long")
    custom = None
    if args.profile == "custom-long":
        from simmr_amd import CustomShortErrorProfile, model_io
        lm, ls = (float(x) for x in args.length_normal.split(","))
        custom = CustomShortErrorProfile(model_io.synthetic_long_model(kmer_size=7, n_positions=1000, seed=1, n_kmers=4 ** 7,
                                                                       read_length_mean=lm, read_length_std=ls))
        prof = custom.pod()  # `custom` owns the model bytes the POD points to
        prof.length_mode = _abi.LEN_PER_READ
        prof.long_start_mode = _abi.START_UNIFORM
        # --rng philox (the default): the draws of the k-mer splice from Philox counters; qualities, lengths and positions are
        # the reference mode's in both (include/simmr_hip.h, enum simmr_rng_mode)
    elif args.profile == "custom-short":
        from simmr_amd import CustomShortErrorProfile, model_io
        custom = CustomShortErrorProfile(model_io.synthetic_short_model())  # 120 modelled positions, lengths ~ N(140, 12)
        prof = custom.pod()
        args.rng = "reference"
        args.no_other_mode = True
    elif long_mode:
        gm, gs = (float(x) for x in args.gamma.split(","))
        prof = MinimalLongErrorProfile(gamma_mean=gm, gamma_std=gs, length_mode=_abi.LEN_PER_READ).pod()
    else:
        prof = (MinimalShortErrorProfile() if args.profile == "minimal-short" else PerfectShortErrorProfile()).pod()
    if args.rng == "philox" and args.profile != "perfect-short":
        prof.rng_mode = _abi.RNG_PHILOX
        if args.plan == "philox" and args.profile in ("minimal-short", "minimal-long"):
            prof.rng_mode = _abi.RNG_PHILOX_FULL
    plan_full = prof.rng_mode == _abi.RNG_PHILOX_FULL

    # default layout: the 16-byte read slots where the emit kernel offers them (counter mode, minimal profiles), else compact
    if args.layout is None:
        args.layout = "slot16" if (prof.rng_mode != _abi.RNG_REFERENCE and custom is None and args.profile != "perfect-short") else "compact"
    slot16 = args.layout == "slot16"
    if slot16 and (prof.rng_mode == _abi.RNG_REFERENCE or custom is not None):
        raise SystemExit("--layout slot16 is the counter mode's layout (minimal-short / minimal-long with --rng philox)")
    eng.set_read_slots(16 if slot16 else 0)
    eng.set_plan_overlap(args.plan_overlap)

    pairs_per_gpu = args.reads // 2
    total_reads = 2 * pairs_per_gpu * world  # the whole job
    first = rank * (2 * pairs_per_gpu if long_mode else pairs_per_gpu)
    counters_dev = torch.zeros(_abi.N_COUNTERS, dtype=torch.int64, device=eng.device)

    # Each rank summarizes the outer-stream slots of its own pair range once per step and the ranks
    # exchange 4 numbers, so that no rank re-walks the stream from slot 0 (simulate.seek_outer_stream).
    from simmr_amd.simulate import seek_outer_stream
    pieces = [(0, 1, j * pairs_per_gpu, (j + 1) * pairs_per_gpu) if j + 1 < world else None for j in range(world)]

    # The position of the outer stream at the start of this rank's shard is a pure function of (seed, shard): it is
    # found once, before the timed steps (one exchange of 4 numbers per rank), and every step plans from it — a run
    # seeks once, however many shards of output it then produces.
    shard_start = {}

    def plan():
        if long_mode:  # shard = range of global read indices
            return eng.long_plan([0], [total_reads], prof, args.seed, first, 2 * pairs_per_gpu)
        if "pos" not in shard_start:  # (SIMMR_RNG_PHILOX_FULL: a pair's draws are a function of its index, no stream to seek in)
            shard_start["pos"] = seek_outer_stream(eng, pieces, (0, 1, first), args.seed) if (world > 1 and prof.rng_mode != _abi.RNG_PHILOX_FULL) else (0, 0)
        return eng.pe_plan(0, prof, total_reads, args.seed, first, pairs_per_gpu, shard_start["pos"])

    # sizes are a deterministic function of (seed, shard): plan once to allocate
    info = plan()
    # (one pair of buffers serves both layouts of the side measurements: sized for the larger, the slots)
    slot_capable = prof.rng_mode != _abi.RNG_REFERENCE and custom is None and args.profile != "perfect-short"
    cap_bases = info.total_bases
    if slot_capable and not slot16 and world == 1 and not args.no_other_mode:
        eng.set_read_slots(16)
        cap_bases = max(cap_bases, plan().total_bases)
        eng.set_read_slots(0)
        info = plan()
    if world == 1 and not args.no_other_mode and args.rng == "philox" and args.profile in ("minimal-short", "minimal-long"):
        # (the `other_plan` side measurement draws other lengths: size the buffers for the larger of the two runs)
        keep_mode = prof.rng_mode
        prof.rng_mode = _abi.RNG_PHILOX if plan_full else _abi.RNG_PHILOX_FULL
        for sb in ((16, 0) if slot_capable else (16 if slot16 else 0,)):
            eng.set_read_slots(sb)
            cap_bases = max(cap_bases, plan().total_bases)
        prof.rng_mode = keep_mode
        eng.set_read_slots(16 if slot16 else 0)
        info = plan()
    out = Reads.allocate(info.n_reads, cap_bases, eng.device, qual_offset=33, slot_bytes=info.slot_bytes)
    out.total_bases = int(info.total_bases)
    out_main = out
    # the reference's default read header (cli.rs:193-200) and ids for the synthetic genome
    FQ_FMT = ("@{:read_id:}|{:genome_id:}/{:pair:} metadata:sid={:sequence_id:}|sp={:start_position:}"
              "|ep={:end_position:}|rc={:reverse_complement:}")
    FQ_NAMES = [(0, "0f8fad5b-d9cb-469f-a165-70867728950e", ["synth_%d" % args.genome_bases])]
    fq = {"text": None, "bytes": 0, "plan_ms": [], "emit_ms": []}

    def step_fastq(record):
        eng.counters_reset()
        plan()
        total = eng.fastq_plan_direct(FQ_FMT, FQ_NAMES, 0)
        if fq["text"] is None or fq["text"].numel() < total:
            fq["text"] = torch.empty(total, dtype=torch.uint8, device=eng.device)
        fq["bytes"] = total
        eng.emit_fastq(fq["text"])
        eng.counters_to(counters_dev)
        if world > 1:
            if args.backend == "nccl":
                dist.all_reduce(counters_dev)
            else:
                c = counters_dev.cpu()
                dist.all_reduce(c)
                counters_dev.copy_(c)
        if record:  # (the emit kernel's time is read once after the timed steps: asking after every emit would wait for it)
            plan_ms.append(eng.last_plan_ms())
            fq["plan_ms"].append(eng.last_fastq_plan_ms())

    emit_ms, plan_ms = [], []

    def step(record):
        eng.counters_reset()
        out.total_bases = int(plan().total_bases)
        if long_mode:
            eng.long_emit(0, out)
        else:
            eng.pe_emit(0, out)  # read ids: pair 0 of the (only) genome is id 0
        eng.counters_to(counters_dev)
        if world > 1:  # the path's only collective: RCCL over xGMI
            if args.backend == "nccl":
                dist.all_reduce(counters_dev)
            else:
                c = counters_dev.cpu()
                dist.all_reduce(c)
                counters_dev.copy_(c)
        if record:  # (the emit kernel's time is read once after the timed steps: asking after every emit would wait for it
            #  and keep the plan of the next step from running beside it)
            plan_ms.append(eng.last_plan_ms())

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    the_step = step_fastq if args.through_fastq else step
    if args.through_fastq:
        step(False)  # (untimed) the columns once, for the read lengths behind `alg_bytes_per_launch`
    for _ in range(args.warmup):
        the_step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        the_step(True)
    fence()
    elapsed = time.perf_counter() - t0
    # HIP events around every emit launch of the timed steps, on the engine's stream (the last 64 are kept)
    emit_ms.append(eng.emit_kernel_ms_mean(min(args.steps, 64)))
    el = torch.tensor([elapsed], dtype=torch.float64, device=eng.device if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    # untimed side measurement of the other rng mode on the same shard (N = 1 only)
    other = None
    if world == 1 and not args.no_other_mode and args.profile != "perfect-short":
        keep_mode, keep_c = prof.rng_mode, counters_dev.clone()
        prof.rng_mode = _abi.RNG_REFERENCE if args.rng == "philox" else (_abi.RNG_PHILOX_FULL if plan_full else _abi.RNG_PHILOX)
        if slot16:  # (the reference's streams are walked by kernels that write the compact layout)
            eng.set_read_slots(0)
            out.slot_bytes = 0
        step(False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        kms = eng.last_emit_kernel_ms()
        other = {
            "rng": "reference StdRng streams (ChaCha12), bit-exact vs the reference algorithm" if args.rng == "philox"
                   else "Philox4x32-10 counter mode (tolerance parity)",
            "value": info.n_reads / dt, "unit": "reads/s", "ms_per_step": dt * 1e3,
            "kernel": ("k_custom_long_qual + k_custom_long_splice" + ("" if args.rng == "philox" else "<CTR>")) if custom is not None
                      else "k_emit_lanes" if args.rng == "philox" else "k_emit_philox", "kernel_ms": kms,
            "steps": 1, "note": "one step after one warm-up step, same shard, outside the timed region",
        }
        prof.rng_mode = keep_mode
        counters_dev.copy_(keep_c)
        if slot16:
            eng.set_read_slots(16)
            out.slot_bytes = 16

    # untimed side measurement of the other output layout (N = 1, counter mode only)
    other_layout = None
    if world == 1 and not args.no_other_mode and slot_capable and not args.through_fastq:
        keep_c = counters_dev.clone()
        eng.set_read_slots(0 if slot16 else 16)
        out.slot_bytes = 0 if slot16 else 16
        step(False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step(False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        other_layout = {"layout": "compact" if slot16 else "slot16", "value": info.n_reads / dt, "unit": "reads/s",
                        "ms_per_step": dt * 1e3, "kernel": "k_emit_philox", "kernel_ms": eng.last_emit_kernel_ms(),
                        "stream_bytes": 2 * out.total_bases,
                        "steps": 1, "note": "one step after one warm-up step, same shard, outside the timed region"}
        eng.set_read_slots(16 if slot16 else 0)
        out.slot_bytes = 16 if slot16 else 0
        step(False)  # (the columns of the timed layout again: the read lengths below come from them)
        counters_dev.copy_(keep_c)

    # untimed side measurement of the other plan (N = 1): the same step with the plan drawn from the reference's streams
    # (SIMMR_RNG_PHILOX: what `value` was measured with before the full counter mode existed) — or from counters
    other_plan = None
    if world == 1 and not args.no_other_mode and args.rng == "philox" and args.profile in ("minimal-short", "minimal-long") \
            and not args.through_fastq:
        keep_mode, keep_c = prof.rng_mode, counters_dev.clone()
        prof.rng_mode = _abi.RNG_PHILOX if plan_full else _abi.RNG_PHILOX_FULL
        step(False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(3):
            step(False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / 3
        other_plan = {"plan": ("the reference's ChaCha12 streams (SIMMR_RNG_PHILOX: positions, lengths and seeds of the reference run)"
                               if plan_full else "Philox counters (SIMMR_RNG_PHILOX_FULL)"),
                      "value": info.n_reads / dt, "unit": "reads/s", "ms_per_step": dt * 1e3, "plan_ms": eng.last_plan_ms(),
                      "kernel_ms": eng.last_emit_kernel_ms(), "steps": 3,
                      "note": "three steps after one warm-up step, same shard, outside the timed region"}
        prof.rng_mode = keep_mode
        step(False)  # (the columns of the timed mode again: the read lengths below come from them)
        counters_dev.copy_(keep_c)

    # untimed side measurement (N = 1): the same steps with the plan of step k + 1 beside the emit of step k
    with_overlap = None
    if world == 1 and not args.no_other_mode and not args.no_overlap_side and not args.plan_overlap and not args.through_fastq:
        keep_c = counters_dev.clone()
        eng.set_plan_overlap(True)
        for _ in range(2):
            step(False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            step(False)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / 10
        with_overlap = {"what": "simmr_engine_set_plan_overlap(e, 1): plan calls on the engine's own stream into a second set of plan "
                                "buffers — the plan of step k + 1 runs while the emit of step k is on the device",
                        "value": info.n_reads / dt, "unit": "reads/s", "ms_per_step": dt * 1e3,
                        "kernel_ms": eng.emit_kernel_ms_mean(10), "steps": 10,
                        "note": "ten steps after two warm-up steps, same shard, outside the timed region; the emit kernel shares the "
                                "device with the plan kernels here, so its time is not the kernel's own"}
        eng.set_plan_overlap(False)
        step(False)
        counters_dev.copy_(keep_c)

    # untimed side measurement: the same step through FASTQ text (N = 1, the default command only)
    through = None
    if world == 1 and not args.no_other_mode and not args.through_fastq and not long_mode and custom is None:
        keep_c = counters_dev.clone()
        step_fastq(False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        step_fastq(False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        through = {"what": "plan + FASTQ sizing + emit into FASTQ text (simmr_fastq_plan_direct / simmr_emit_fastq), text resident in HBM",
                   "value": info.n_reads / dt, "unit": "reads/s", "ms_per_step": dt * 1e3, "text_bytes": fq["bytes"],
                   "emit_kernel_ms": eng.last_emit_kernel_ms(), "plan_ms": eng.last_plan_ms(), "fastq_plan_ms": eng.last_fastq_plan_ms(),
                   "steps": 1, "note": "one step after one warm-up step, same shard, outside the timed region"}
        fq["text"] = None
        counters_dev.copy_(keep_c)

    counters = counters_dev.cpu().numpy().astype(np.uint64)
    n_reads_job = int(counters[_abi.CNT_READS])
    n_bases_job = int(counters[_abi.CNT_BASES])
    assert n_reads_job == 2 * pairs_per_gpu * world, (n_reads_job, pairs_per_gpu, world)

    # algorithmic bytes of ONE emit launch on this rank:
    # per read ceil(L/4) packed-reference bytes + L bases + L qualities + 16 metadata
    # (the padding of the slot layout is traffic, not credit: bases and qualities count L bytes each in either layout)
    lens = ((out.end[:info.n_reads] - out.start[:info.n_reads]).abs() if slot16 else
            out.seq_off[1:info.n_reads + 1] - out.seq_off[:info.n_reads])
    plane_bytes = int(((lens + 3) // 4).sum().item())
    alg_bytes = plane_bytes + 2 * int(lens.sum().item()) + 16 * int(info.n_reads)
    # ... and of one launch of the TEXT form (simmr_emit_fastq): the packed-reference bytes read + every byte of the FASTQ
    # text written (headers, bases, "+" lines, qualities, line ends: fastq.rs:58-66); the text replaces the SoA columns
    text_alg_bytes = plane_bytes + int(fq["bytes"])
    if through is not None and through["emit_kernel_ms"] > 0:
        a = text_alg_bytes / (through["emit_kernel_ms"] * 1e-3) / 1e9
        through["roofline"] = {"bound": "hbm", "kernel": text_kernel_label(args),
                               "achieved": a, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": a / HBM_PEAK_GBPS,
                               "traffic": measured_traffic(args, 2 * pairs_per_gpu, text=True),
                               "alg_bytes_per_launch": text_alg_bytes, "kernel_ms": through["emit_kernel_ms"],
                               "note": "algorithmic bytes = FASTQ text written + 2-bit reference read; one launch, HIP events on the engine's stream"}
    if args.through_fastq:
        alg_bytes = text_alg_bytes
    if other_layout is not None and other_layout["kernel_ms"] > 0:
        other_layout["achieved_GBps"] = alg_bytes / (other_layout["kernel_ms"] * 1e-3) / 1e9
        other_layout["roofline_frac"] = other_layout["achieved_GBps"] / HBM_PEAK_GBPS
    if other is not None and other["kernel_ms"] > 0:
        other["achieved_GBps"] = alg_bytes / (other["kernel_ms"] * 1e-3) / 1e9
        other["roofline_frac"] = other["achieved_GBps"] / HBM_PEAK_GBPS
    emit_avg_ms = sum(emit_ms) / max(len(emit_ms), 1)
    achieved = alg_bytes / (emit_avg_ms * 1e-3) / 1e9 if emit_avg_ms > 0 else 0.0

    result = None
    if rank == 0:
        value = n_reads_job * args.steps / elapsed
        subst_rate = float(counters[_abi.CNT_SUBSTITUTIONS]) / max(float(counters[_abi.CNT_ACGT_BASES]), 1.0)
        result = {
            "metric": "simulated_reads_per_sec",
            "value": value,
            "unit": "reads/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic",
            "config": {
                "workload": ("custom-short (synthetic simmrd-shaped model: 120 modelled positions, lengths ~ N(140, 12), inserts ~ N(200, 40)) PE"
                             if args.profile == "custom-short" else
                             f"custom-long (synthetic simmrd-shaped model, k = 7, N({args.length_normal}) per-read lengths, uniform starts)"
                             if custom is not None else
                             f"{args.profile} gamma({args.gamma}) long reads" if long_mode else f"{args.profile} 150 bp PE")
                            + f", 1 genome ({args.genome_bases} bp synthetic SplitMix64 seed 2), "
                            f"{2 * pairs_per_gpu} reads per GPU per step, seed {args.seed}",
                "rng": ("reference StdRng streams (ChaCha12), bit-exact mode" if args.rng == "reference" else
                        "Philox4x32-10 counter mode for every draw of the path — contig, seeds, lengths, positions and the per-base "
                        "draws (SIMMR_RNG_PHILOX_FULL, include/simmr_hip.h; tolerance parity)" if plan_full else
                        "Philox4x32-10 counter mode for per-base draws, plan from the reference's streams (tolerance parity)"),
                "layout": ("slot16: every read in a 16-byte-aligned slot of seq / qual (SIMMR_SLOT16, include/simmr_hip.h), "
                           f"{2 * int(info.total_bases)} stream bytes per step" if slot16 else
                           f"compact: seq / qual byte streams without gaps, {2 * int(info.total_bases)} stream bytes per step"),
                "plan_overlap": ("off" if not args.plan_overlap else
                                 "on: the plan of step k + 1 runs on the engine's plan stream beside the emit of step k "
                                 "(simmr_engine_set_plan_overlap; every step still plans and emits its own shard)"),
                "reads_per_gpu": 2 * pairs_per_gpu,
                "sharding": ("global read-index range per GPU" if long_mode else "pair-index range per GPU"),
                # what carried the collectives of this run, and how many ranks it saw
                "backend": ("none (one rank)" if world == 1 else
                            "nccl (RCCL)" if args.backend == "nccl" else "gloo (CPU rehearsal: every rank on one device)"),
                "world_size_seen": dist.get_world_size() if world > 1 else 1,
                "collectives": ("none" if world == 1 else
                                "per step one all-reduce of %d run counters; once before the timed steps one all-gather of 4 x i64 "
                                "per rank (outer-stream seek)" % _abi.N_COUNTERS),
            },
            "gbases_per_sec": n_bases_job * args.steps / elapsed / 1e9,
            "substitution_rate": subst_rate,
            "mean_phred": float(counters[_abi.CNT_QUAL_SUM]) / max(float(counters[_abi.CNT_BASES]), 1.0),
            "plan_ms_per_step": sum(plan_ms) / max(len(plan_ms), 1),
            "roofline": {
                # "hbm" is the roofline the fraction below is taken against (the contract's two choices are hbm and
                # mfma, and this path has no matrix work); what actually limits the kernel is in `limiter`
                "bound": "hbm",
                "limiter": ("hbm write path" if args.profile == "perfect-short" else
                            "per-lane table reads (address unit)" if args.profile == "custom-short" else
                            "latency of the dependent table load per visited k-mer at 4 waves per SIMD, and ChaCha12 issue"
                            if (custom is not None and args.rng == "reference") else
                            "latency of the sequential k-mer walk at six waves per SIMD (one lane per read; per base an LDS lookup, "
                            "the level-1 test, and for one lane in five a column load), with VALU issue (45 instructions per base) "
                            "at half of the kernel's time"
                            if custom is not None else
                            "VALU issue (ChaCha12 and the per-base state machines of the reference's streams)"
                            if args.rng == "reference" else
                            "VALU issue (integer RNG and table lookups) next to the issue of the store instructions and the "
                            "HBM write path behind them: see `valu` and DESIGN.md section 4"),
                "kernel": (text_kernel_label(args) if args.through_fastq else
                           "k_emit_perfect_pe" if args.profile == "perfect-short" else
                           "k_emit_custom_pe" if args.profile == "custom-short" else
                           "k_custom_long_qual + k_custom_long_splice" + ("<CTR>" if args.rng == "philox" else "") if custom is not None else
                           "k_emit_lanes" if args.rng == "reference" else "k_emit_philox"),
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": measured_traffic(args, 2 * pairs_per_gpu),
                "alg_bytes_per_launch": alg_bytes,
                "kernel_ms": emit_avg_ms,
                "note": ("HBM-write bound data movement" if args.profile == "perfect-short" else
                         "see DESIGN.md section 4 (kernel table)" if (custom is not None or args.profile == "custom-short"
                                                                     or args.rng == "reference") else
                         "see DESIGN.md section 4: VALU issue is the first limiter (class-priced share in `valu`), the store path "
                         "next to it; 16-byte read slots, nontemporal whole-line stores, 128 workgroups per CU over sharded run "
                         "counters; LAB.md has the history"),
            },
        }
        valu = measured_valu(args, 2 * pairs_per_gpu, emit_avg_ms)
        if valu is not None:
            result["roofline"]["valu"] = valu
        if other is not None:
            result["other_rng_mode"] = other
        if other_layout is not None:
            result["other_layout"] = other_layout
        if other_plan is not None:
            result["other_plan"] = other_plan
        if with_overlap is not None:
            result["with_plan_overlap"] = with_overlap
        if through is not None:
            result["through_fastq"] = through
        if args.through_fastq:
            result["config"]["output"] = "FASTQ text in HBM (header format of cli.rs:193-200), no SoA columns"
            result["fastq"] = {"text_bytes": fq["bytes"], "fastq_plan_ms_per_step": sum(fq["plan_ms"]) / max(len(fq["plan_ms"]), 1),
                               "text_GBps": fq["bytes"] * args.steps / elapsed / 1e9}
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the CPU leg runs on rank 0 after the ranks have parted (nobody waits in a collective while it runs)
        result["reference_toolchain"] = reference_toolchain()
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline_long(args, prof) if long_mode else cpu_baseline(args, prof)
        print(json.dumps(result), flush=True)


def text_kernel_label(args):
    """which kernel simmr_emit_fastq runs for this command: the item form, or (SIMMR_TEXT_FORM=2, read by the engine when it
    is made) the whole-line form of simmr_amd/csrc/text_lines.hip for paired plans of short reads"""
    if os.environ.get("SIMMR_TEXT_FORM") == "2" and args.profile in ("minimal-short", "perfect-short"):
        return "k_emit_text_lines<COPY_ONLY>" if args.profile == "perfect-short" else "k_emit_text_lines"
    return "k_emit_philox<COPY_ONLY, TEXT>" if args.profile == "perfect-short" else "k_emit_philox<TEXT>"


PMC_RECORD = ROOT / "profiles" / "r5" / "pmc_traffic.json"
# everything the counters of a launch depend on: the kernels, and the launch geometry and kernel selection in engine.hip
KERNEL_SOURCES = ("simmr_amd/csrc/kernels.hip", "simmr_amd/csrc/rng_device.hpp", "simmr_amd/csrc/device_types.hpp",
                  "simmr_amd/csrc/fastq_format.hpp", "simmr_amd/csrc/fastq_kernels.hip", "simmr_amd/csrc/text_lines.hip",
                  "simmr_amd/csrc/engine.hip")


def kernel_source_hash():
    """sha256 over the sources that define the emit kernels: the committed PMC record carries the hash of the sources it
    was collected from (tools/make_pmc_traffic.py), and is only reported while they are unchanged."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        h.update((ROOT / rel).read_bytes())
    return h.hexdigest()


def _profile_record(args, reads_per_gpu, text=False):
    """(record, why_not): the committed rocprofv3 PMC record of this very command (profiles/r5/pmc_traffic.json: one
    counter set per run, no tracing) — only for the workload it was measured on and only if the kernel sources still
    hash to what they were when it was collected."""
    try:
        t = json.load(open(PMC_RECORD))
    except OSError:
        return None, f"no {PMC_RECORD.relative_to(ROOT)}"
    if t.get("source_sha256") != kernel_source_hash():
        return None, (f"{PMC_RECORD.relative_to(ROOT)} was collected from other kernel sources "
                      f"(sha256 {str(t.get('source_sha256'))[:12]}.. != {kernel_source_hash()[:12]}..): collect it again")
    if args.through_fastq or text:
        if reads_per_gpu == 100_000_000 and args.genome_bases == 100_000_000 and args.profile == "minimal-short" and args.rng == "philox":
            return (t.get("k_emit_philox_text"), None) if t.get("k_emit_philox_text") else (None, "not collected for k_emit_philox_text")
        return None, "the TEXT form's counters were collected for the default workload only"
    if args.profile == "custom-long" and reads_per_gpu == 1_000_000 and args.genome_bases == 100_000_000:
        key = "k_custom_long_splice" if args.rng == "reference" else "k_custom_long_splice_ctr"
        return (t.get(key), None) if t.get(key) else (None, f"not collected for {key}")
    if reads_per_gpu != 100_000_000 or args.genome_bases != 100_000_000:
        return None, "collected at 100 M reads on 100 Mbp only"
    if args.profile == "minimal-short" and args.rng == "philox":
        key = "k_emit_philox_slot16" if args.layout == "slot16" else "k_emit_philox"
        return (t.get(key), None) if t.get(key) else (None, f"not collected for {key}")
    if args.profile == "perfect-short":
        return t.get("k_emit_perfect_pe"), None
    return None, "not collected for this profile"


def measured_traffic(args, reads_per_gpu, text=False):
    """HBM bytes per launch of the dominant kernel: FETCH_SIZE + WRITE_SIZE of separate --pmc passes, raw (the guide's
    x2 correction of FETCH_SIZE applies to wide coalesced reads; this kernel reads 8-byte gathers and plan columns, so
    the raw figure is reported and the doubled one is in the file).  Not measured by this run: a constant from the
    committed profile, null for any other workload or when the kernel sources have changed since."""
    t, _ = _profile_record(args, reads_per_gpu, text)
    return None if t is None else t.get("bytes_raw")


def measured_valu(args, reads_per_gpu, kernel_ms):
    """VALU wave-instructions per launch (SQ_INSTS_VALU of the committed PMC pass) priced by instruction class with the
    measured issue rates (profiles/microbench/valu_asm_rates*_mi355x.txt: plain 32-bit forms 8.4e11 wave-instructions/s
    on the whole chip, the VOP3 / SDWA / DPP class 1.5 x slower, v_mad_u64_u32 2.1 x): the time the instruction stream
    alone needs, as a share of the kernel's time."""
    t, why = _profile_record(args, reads_per_gpu)
    if t is None or "SQ_INSTS_VALU" not in t or kernel_ms <= 0:
        return {"note": why} if why else None
    peak = 8.443e11
    need_ms = t["SQ_INSTS_VALU"] / peak * 1e3
    out = {"wave_instructions_per_launch": t["SQ_INSTS_VALU"], "issue_peak_per_s": peak, "unit": "wave-instructions",
           "min_time_ms_at_peak": need_ms, "frac_of_kernel_time": need_ms / kernel_ms,
           "source": f"{PMC_RECORD.relative_to(ROOT)} (committed PMC pass of this command, kernel sources unchanged since), "
                     "not measured by this run"}
    mix = t.get("valu_class_mix")  # shares of the instruction stream by issue class, from the disassembly
    if mix:
        cost = mix.get("plain", 0.0) * 1.0 + mix.get("vop3_sdwa", 0.0) * 1.5 + mix.get("mad_u64", 0.0) * 2.1
        out["class_priced_min_time_ms"] = need_ms * cost
        out["class_priced_frac_of_kernel_time"] = need_ms * cost / kernel_ms
        out["class_mix"] = mix
    return out


def usable_cores():
    """CPU share of this process: the cgroup quota when there is one (a GPU box gives one
    GPU's job 16 CPUs although 256 hardware threads are visible), else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(round(int(txt[0]) / int(txt[1])))))
            else:
                q = int(txt[0])
                if q > 0:
                    n = min(n, max(1, int(round(q / int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())))))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def reference_toolchain():
    """the reference is Rust; say whether this box could have built it."""
    import shutil
    import subprocess
    exe = shutil.which("cargo")
    if exe is None:
        return "absent (no cargo on PATH: the reference cannot be built or timed here; cpu_baseline is the C restatement)"
    try:
        return subprocess.run([exe, "--version"], capture_output=True, text=True, timeout=20).stdout.strip() or "cargo present, no version"
    except (OSError, subprocess.SubprocessError) as e:
        return f"cargo at {exe} does not run ({e.__class__.__name__})"


def _reference_mode_copy(prof):
    import ctypes
    from simmr_amd import _abi
    ref = type(prof)()
    ctypes.memmove(ctypes.byref(ref), ctypes.byref(prof), ctypes.sizeof(prof))
    ref.rng_mode = _abi.RNG_REFERENCE
    return ref


def cpu_baseline_long(args, prof):
    """BASELINE.md section 3, run B4: the CPU oracle's simulate_long_reads (simulate.rs:323-406,
    minimal_long.rs:58-73 / custom_short.rs for the model) on the first reads of the same run, one thread and
    all of this host's, with the reference's generator.  The default figure leaves out what the reference does
    on top of the algorithm — a clone of every usable sequence of the genome plus one of the chosen sequence
    per read (simulate.rs:362-375) — and `faithful_cost` times a smaller sample with those copies made."""
    from tests import _oracle, _synth
    lib = _oracle.load()
    prof = _reference_mode_copy(prof)
    genome = _oracle.HostGenome(_synth.synthetic_contigs([args.genome_bases], 2))
    cores = usable_cores()
    total_reads = 2 * (args.reads // 2) * args.gpus

    def timed(n, threads, faithful=False):
        lib.orc_set_faithful_cost(1 if faithful else 0)
        try:
            _oracle.simulate_long(lib, [genome], [total_reads], prof, args.seed, 0, min(n, 64), threads=threads)  # pages, tables
            t = time.perf_counter()
            o = _oracle.simulate_long(lib, [genome], [total_reads], prof, args.seed, 0, n, threads=threads)
            return time.perf_counter() - t, o
        finally:
            lib.orc_set_faithful_cost(0)
    # sized for about 10-30 s of CPU work in all: the model's k-mer splice builds an alias table per visited k-mer
    # (custom_short.rs:497-500), which makes a custom long read cost 0.2 s on one thread
    custom = args.profile == "custom-long"
    n = min(args.cpu_sample_long_reads or (1000 if custom else 400_000), total_reads)
    n1 = max(n // (16 if custom else 8), 1)
    t1, _ = timed(n1, 1)
    tn, o = timed(n, cores)
    nf = max(min(n // (50 if custom else 1000), 60), 1)
    tf, of = timed(nf, 1, faithful=True)
    return {
        "value": n / tn, "unit": "reads/s", "cores": cores, "kind": "port",
        "sample": f"first {n} long reads of the same run (same genome, profile, seed; the reference's ChaCha12 streams), OpenMP over "
                  f"reads on {cores} threads, {tn:.1f} s; without the reference's per-read clones of the genome (simulate.rs:362-375)",
        "gbases_per_sec": o.total_bases / tn / 1e9,
        "single_thread_value": n1 / t1, "single_thread_gbases_per_sec": _bases_of(o, n1) / t1 / 1e9,
        "single_thread_sample": f"first {n1} reads, 1 thread, {t1:.1f} s",
        "faithful_cost": {"value": nf / tf, "unit": "reads/s", "cores": 1, "gbases_per_sec": of.total_bases / tf / 1e9,
                          "sample": f"first {nf} reads, 1 thread, {tf:.1f} s, with the clones of simulate.rs:362-375 made "
                                    f"(every usable sequence once, the chosen one again: {2 * args.genome_bases} bytes per read)"},
    }


def _bases_of(o, n):
    return int(o.seq_off[n])


def cpu_baseline(args, prof):
    """The CPU oracle — a port of the reference algorithm with the reference's own
    generator (the Rust reference cannot be built here) — on a bounded sample of the
    same workload, on this host's cores.  Output buffers are allocated and touched
    before the clock starts."""
    import numpy as np
    from simmr_amd import _abi
    from tests import _oracle, _synth
    lib = _oracle.load()
    prof = _reference_mode_copy(prof)
    contigs = _synth.synthetic_contigs([args.genome_bases], 2)
    genome = _oracle.HostGenome(contigs)
    cores = usable_cores()

    def timed(n, threads):
        out = _oracle.HostReads(n, n * 176, 0)
        for a in (out.seq, out.qual, out.seq_off, out.start, out.end, out.contig, out.genome, out.read_id, out.flags):
            a.fill(0)  # fault the pages in now
        t = time.perf_counter()
        o = _oracle.simulate_pe(lib, genome, prof, n, args.seed, threads=threads, out=out)
        return time.perf_counter() - t, o
    # about 10-30 s of CPU work in all (an empirical-model pair costs 100 x a minimal-short one on the CPU)
    n = args.cpu_sample_reads or (400_000 if args.profile == "custom-short" else 20_000_000)
    n = min(n, 2 * (args.reads // 2) * args.gpus)
    n1 = max(2, min(n // 8, 20_000 if args.profile == "custom-short" else 250_000))
    t1, _ = timed(n1, 1)
    tn, o = timed(n, cores)
    return {
        "value": n / tn,
        "unit": "reads/s",
        "cores": cores,
        "kind": "port",
        "sample": f"first {n} reads of the same run (same genome, profile, seed; the reference's ChaCha12 streams), "
                  f"OpenMP over pairs on {cores} threads, {tn:.1f} s",
        "single_thread_value": n1 / t1,
        "single_thread_sample": f"first {n1} reads, 1 thread, {t1:.1f} s",
        "gbases_per_sec": o.total_bases / tn / 1e9,
    }


if __name__ == "__main__":
    main()
