#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X: "BLS12-381 G1 scalar-muls/sec at MSM size 2^20; shuffle proofs verified/sec".

    python bench.py --gpus 1 --steps 10 --warmup 2
    python bench.py --gpus N --steps K --warmup W          # spawns its own N ranks (fresh child processes, one per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W               # or under a launcher that exports RANK / LOCAL_RANK / WORLD_SIZE
No PyTorch is imported on any path: the N>1 exchange is the library's own (cg1_comm_*: RCCL all-gather of one 144-byte
partial G1 sum per rank over xGMI, TCP control channel for rendezvous / barriers / the max-over-ranks clock).

HEADLINE (`value`): G1 scalar-muls/s.  A "step" is one complete compute_MSM-equivalent call (msm_accumulator.py:6-12 of
the reference) over one batch of synthetic input that is already resident in HBM: point preparation, signed-digit recode,
counting sort, bucket accumulation, bucket reduction, D2H of the window sums and the host Horner tail are all inside the
timed region.  Workload at N=1: BASELINE.json configs[1]'s shape at the metric's size -- one MSM of 2^20 random G1 points
(k_i*G) with scalars uniform in [1, r-1] (the reference's random_scalar, util.py:21-24).  At N>1 (weak scaling): ONE MSM of
N*2^20 terms.  --shard hybrid (default): the signed-digit windows are sharded over 2 window-bucket groups and the points over
N/2 point groups (rank = window group + 2 * point group); --shard windows: pure window sharding (every rank holds all
points, what north_star names; also timed as `windows_only` in every N>1 line); --shard points: pure point sharding.  The
partial G1 sums are all-gathered over RCCL (cg1_comm_allreduce_g1) and added on every rank.  value = total terms / max-over-ranks time.

SECONDARY (same JSON line, key `secondary`, N=1): the metric's second half -- Whisk shuffle proofs verified/s
(IsValidWhiskShuffleProof, whisk_interface.py:72-109 -> curdleproofs.py:162-248; BASELINE configs[2]): batches of 1024
proofs (the 1024 DISTINCT ell = 124 proofs of tests/golden/shuffle_batch_ell124{,_more}.bin, made by the reference prover,
fresh random weights per slot) from wire bytes in host memory to verdicts, with per-phase times, its own cpu_baseline and the
integer-MAD roofline of its dominant kernel k_batch_decompress.  `--mode verify` runs that part alone (and proof-per-GPU at N>1:
BASELINE configs[4]'s structure).
"""
import argparse
import json
import os
import statistics
import sys
import time

# The application's choice, made before anything initialises HIP (curdleproofs_pie_amd._native.tune_runtime does the same): the verifier's
# three pipelines keep a dozen streams busy and the runtime's default is 4 hardware queues.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MAD_PEAK_T = 30.3          # chip-wide v_mad_u64_u32 rate, T lane-ops/s, round 1 on another box (profiles/r01_ubench_valu_rates.txt): kept as a
                           # reference only -- every run measures its own peak (mad_peak_same_run) and prices the roofline against that
FR_ORDER = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
MADS_PER_MADD = 3542       # XYZZ mixed add (g1_xyzz.h): 6 products x 392 + one fused double product x 588 + 2 squarings x 301
MADS_PER_MMADD = 1974      # affine + affine (xyzz_mmadd, the first addition of a chunk): 2 x 392 + 588 + 2 x 301
MADS_MUL, MADS_SQR = 392, 301
# rank 0's share of ONE MSM of N x 2^20 terms timed on ONE GPU in round 4 (profiles/r04_shard_emulation.txt; that box ran the plain
# single-GPU step in 3.10 ms): what an N-GPU run should show per step before the exchange.  EMULATION, not a multi-GPU measurement.
EMULATED_MS = {"hybrid": {1: 3.10, 2: 3.25, 4: 3.24, 8: 3.26}, "windows": {1: 3.10, 2: 3.25, 4: 3.39, 8: 3.67}, "points": {1: 3.10, 2: 2.97, 4: 2.95, 8: 2.97}}


def raw96_gen():
    gx = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
    gy = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
    return gx.to_bytes(48, "little") + gy.to_bytes(48, "little")


def mad_peak_same_run(ctx):
    """Chip-wide v_mad_u64_u32 issue rate measured in THIS process on THIS box (k_probe_mad_rate, 2 waves per SIMD -- the
    occupancy of k_accumulate).  One ~100 ms launch brings the clock to what the chip holds under this load (launches that
    follow an idle spell measure the ramp: 28 T/s where the loaded chip does 33), then the best of three ~100 ms launches."""
    ctx.probe_mad_rate(2, 200)
    runs = [ctx.probe_mad_rate(2, 200) for _ in range(3)]
    cold = ctx.probe_mad_rate(2, 40)
    return {"peak_T": max(runs) / 1e12, "runs_T": [r / 1e12 for r in runs], "one_20ms_launch_T": cold / 1e12,
            "what": "k_probe_mad_rate: 8 independent v_mad_u64_u32 chains per lane, 2 waves per SIMD, hipEvents on the context's stream; "
                    "peak = best of three ~100 ms launches after a ~100 ms warm-up launch"}


def cpu_baseline(d_points, d_scalars, sample_n):
    """The reference algorithm (naive per-term double-and-add loop) as restated by oracle/msm_oracle.c, 1 thread (the
    reference is single-threaded), on the first `sample_n` terms of the same workload; beside it, clearly labelled
    NON-reference baselines: the textbook bucket method on 1 core and on all host cores (OpenMP)."""
    from oracle import c_oracle as C  # the only use of the oracle in the MSM leg: the reported CPU baseline

    from curdleproofs_pie_amd import _native as N

    p = d_points.download(96 * sample_n)
    s = d_scalars.download(32 * sample_n)
    t0 = time.perf_counter()
    naive = C.compute_msm(p, s, sample_n)
    dt = time.perf_counter() - t0
    t1 = time.perf_counter()
    b1 = C.msm_bucket(p, s, sample_n)
    dt_b = time.perf_counter() - t1
    cores = int(N.cg1_shuffle_default_threads())          # usable CPUs: affinity mask capped by the cgroup quota
    big = min(1 << 18, d_points.nbytes // 96)              # all-cores leg: a larger sample (the bucket method's rate grows with n)
    pb, sb = d_points.download(96 * big), d_scalars.download(32 * big)
    t2 = time.perf_counter()
    bm = C.msm_bucket_mt(pb, sb, big, cores)
    dt_m = time.perf_counter() - t2
    assert naive == b1, "CPU oracle: naive loop and bucket method disagree"
    return {"value": sample_n / dt, "unit": "G1 scalar-muls/s", "cores": 1, "kind": "port",
            "sample": f"naive reference loop (msm_accumulator.py:6-12 restated in C, 255-bit double-and-add + add per term) "
                      f"over the first {sample_n} terms of the same workload, {dt:.1f} s on one host core; "
                      f"cost is linear in n so the 2^20 figure is this rate",
            "stronger_non_reference_baselines": [
                {"what": "textbook bucket-method MSM in C (oracle/msm_oracle.c orc_msm_bucket), same sample (rate grows with n)",
                 "cores": 1, "value": sample_n / dt_b, "seconds": dt_b},
                {"what": f"the same bucket method on all usable host cores (OpenMP, orc_msm_bucket_mt), first {big} terms",
                 "cores": cores, "value": big / dt_m, "seconds": dt_m}]}


# ---------------------------------------------------------------------------------------------- proofs verified / s
def load_batch_fixture():
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from batch_fixture import ShuffleBatch

    return ShuffleBatch()


def decompress_mads_per_point():
    """Algorithmic v_mad_u64_u32 of k_batch_decompress<false> per point (kernels_batch.h): x -> Montgomery (1 product),
    x^3 + 4 (1 squaring + 1 product), y = rhs^((p+1)/4) (fp_sqrt_candidate: its squarings / products are counted from the
    exponent by the same rule the kernel applies), y^2 == rhs (1 squaring + 1 product), y out of Montgomery form (1 product),
    and one more Montgomery conversion for the half of the points whose sign bit asks for p - y."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from sqrt_chain import chain_cost

    nsqr, nmul = chain_cost()
    return (nsqr + 2) * MADS_SQR + (nmul + 4.5) * MADS_MUL, nsqr, nmul


FE_KEY = "front_end (transcript + D / A' + Fr algebra: host threads, or one device launch per batch)"


def verify_measure(ctx, threads, steps, warmup, batch, verify_mode="merged", cpu_leg=True, peak_T=None, front_end="auto"):
    """Stream `steps` batches of `batch` distinct-proof slots through the GPU verifier; returns the `secondary` object."""
    from curdleproofs_pie_amd import _native as N
    from curdleproofs_pie_amd.shuffle_verifier import ShuffleBatchVerifier

    if peak_T is None:
        peak_T = mad_peak_same_run(ctx)["peak_T"]
    fx = load_batch_fixture()
    v = ShuffleBatchVerifier(fx.crs, ctx, threads=threads, device_front_end={"auto": None, "host": False, "device": True}[front_end])
    inst, proofs, want = fx.tiled(batch)
    for _ in range(warmup):
        assert not any(v.verify_packed(inst, proofs, batch, mode=verify_mode))
    if v.device_front_end and warmup:             # every front-end lane builds its tables on first use: untimed, like the warm-up steps
        for st in v.verify_stream(((inst, proofs, batch) for _ in range(v.pipelines * (v.fe_lanes + 2))), mode=verify_mode):
            assert not any(st)
    acc = {}
    ctx.sync()
    t0 = time.perf_counter()
    # K steps as a stream: the three stages of consecutive batches overlap (GPU: decompress k+1 | host: front-end k |
    # GPU: MSM k-1); every batch is verified completely inside the timed region
    for st in v.verify_stream(((inst, proofs, batch) for _ in range(steps)), mode=verify_mode):
        assert not any(st)
        for k, x in v.last_stats.items():
            if k.endswith("_s"):
                acc[k] = acc.get(k, 0.0) + x
    ctx.sync()
    el = time.perf_counter() - t0
    L, C = v.crs.points_per_proof, v.crs.ncrs
    points = batch * L
    # dominant GPU kernel, measured live: hipEvents on the context's stream around launches of k_batch_decompress over the
    # batch's own wire points (the same launch the verifier issues, as one piece)
    wire = N.PinnedBuffer(ctx, points * 48)
    ctx.check(N.cg1_shuffle_gather_points(v.crs.handle, batch, inst, proofs, wire.ptr))
    d_w, d_p, d_s = ctx.alloc(points * 48), ctx.alloc(points * 96), ctx.alloc(points)
    ctx.check(N.cg1_h2d(ctx.handle, d_w.ptr, wire.ptr, points * 48))
    reps = 5
    ctx.check(N.cg1_batch_decompress_device(ctx.handle, d_w.ptr, d_p.ptr, d_s.ptr, points, 0))
    ctx.timer_begin()
    for _ in range(reps):
        ctx.check(N.cg1_batch_decompress_enqueue(ctx.handle, d_w.ptr, d_p.ptr, d_s.ptr, points, 0))
    dec_ms = ctx.timer_end() / reps
    assert not any(d_s.download(points))
    mads_pp, nsqr, nmul = decompress_mads_per_point()
    out = {
        "metric": "shuffle proofs verified/sec (Whisk ell=124+4 blinders, batches of %d, mode %s)" % (batch, verify_mode),
        "value": batch * steps / el, "unit": "proofs/s", "ms_per_step": el / steps * 1e3, "steps": steps, "warmup": warmup,
        "batch": batch, "distinct_proofs": fx.count, "higher_is_better": True,
        "data": "tests/golden/shuffle_batch_ell124{,_more}.bin: %d distinct proofs made by the reference prover over one CRS (cycled when the "
                "batch is larger), fresh OS-random weights per slot; inputs are wire bytes in host memory (H2D included)" % fx.count,
        "points_per_step": points + C, "host_threads": threads,
        "front_end": ("device (k_shuffle_front_end_rows; %d pipeline(s) x %d launches side by side)" % (v.pipelines, v.fe_lanes)) if v.device_front_end else "host",
        "phases_ms_per_step": {"decompress_stage (H2D + k_batch_decompress + D2H, GPU thread)": 1e3 * acc.get("decompress_s", 0) / steps,
                               FE_KEY: 1e3 * acc.get("front_end_s", 0) / steps,
                               "merged_msm (one regime-A MSM of all points, GPU)": 1e3 * acc.get("merged_msm_s", 0) / steps,
                               "note": "the three stages of consecutive batches overlap; they do not add up to ms_per_step"},
        "roofline_int_mad": {"kernel": "k_batch_decompress<false>", "bound": "valu v_mad_u64_u32", "kernel_ms": dec_ms, "points_per_launch": points,
                             "mads_per_point": mads_pp, "sqrt_chain": {"squarings": nsqr, "products": nmul},
                             "achieved": mads_pp * points / (dec_ms * 1e-3) / 1e12, "peak": peak_T, "unit": "T mad/s",
                             "frac": mads_pp * points / (dec_ms * 1e-3) / (peak_T * 1e12), "peak_source": "k_probe_mad_rate in this run"},
    }
    if cpu_leg:
        # CPU port beside it: the same statement builder on ONE core + the statement's MSM by the CPU oracle (bucket method)
        from oracle.shuffle_check import oracle_verdicts
        v1 = ShuffleBatchVerifier(v.crs, ctx, threads=1, device_front_end=False)
        m = 8
        i1, p1, _ = fx.tiled(m)
        t1 = time.perf_counter()
        prep = v1.prepare(i1, p1, m)
        assert oracle_verdicts(v1, prep) == [True] * m
        cpu_dt = (time.perf_counter() - t1) / m
        v1.close()
        out["cpu_baseline"] = {"value": 1.0 / cpu_dt, "unit": "proofs/s", "cores": 1, "kind": "port",
                               "sample": "%d distinct proofs: native front-end on one core + CPU-oracle decode and bucket MSM of each 726-term "
                                         "statement (the reference's own Python verifier over our host C++ backend measured 0.27 s/proof in the "
                                         "build container; it cannot run on the GPU box)" % m}
    for b in (d_w, d_p, d_s, wire):
        b.free()
    v.close()
    return out


def single_proof_measure(ctx, reps=9):
    """Latency of ONE proof through the batch verifiers -- what the reference's IsValidWhiskShuffleProof / IsValidWhiskOpeningProof calls are
    (whisk_interface.py:72-87, 147-169): wire bytes in host memory -> verdict, median of `reps`."""
    from curdleproofs_pie_amd.shuffle_verifier import OpeningBatchVerifier, ShuffleBatchVerifier

    fx = load_batch_fixture()
    v = ShuffleBatchVerifier(fx.crs, ctx)
    inst, proofs, _ = fx.tiled(1)
    assert v.verify_packed(inst, proofs, 1) == [0]
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        st = v.verify_packed(inst, proofs, 1)
        ts.append((time.perf_counter() - t0) * 1e3)
        assert st == [0]
    v.close()
    with open(os.path.join(ROOT, "tests", "golden", "opening_vectors.json")) as f:
        c = json.load(f)["cases"][0]
    ov = OpeningBatchVerifier(ctx)
    item = ((bytes.fromhex(c["r_G"]), bytes.fromhex(c["k_r_G"])), bytes.fromhex(c["k_commitment"]), bytes.fromhex(c["proof"]))
    to = []
    for _ in range(reps):
        t0 = time.perf_counter()
        ok = ov.verify_many([item])
        to.append((time.perf_counter() - t0) * 1e3)
        assert ok == [True]
    return {"shuffle_proof_ell124_ms": statistics.median(ts), "opening_proof_ms": statistics.median(to),
            "what": "ONE proof, wire bytes -> verdict, through ShuffleBatchVerifier / OpeningBatchVerifier with their defaults (an isolated small "
                    "batch takes the host front-end; an opening proof the exact host check)"}


def pmc_traffic(logn):
    """`roofline.traffic`: fabric-side bytes per k_accumulate launch from the last committed rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE in
    their own runs, the gfx950 correction applied: tools/summarize_profiles.py -> profiles/pmc_traffic.json).  Counters cannot be read inside
    this run, so the figure is the collection's, and only for the workload it was taken on (2^20 terms, one launch per step)."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            t = json.load(f)
        if logn != 20 or "2^20" not in t.get("workload", ""):
            raise ValueError("other workload")
        return {"traffic": t["k_accumulate_hbm_bytes_per_launch"], "traffic_unit": "bytes per launch",
                "traffic_source": "profiles/pmc_traffic.json <- " + t.get("source", "?").split(" (")[0] + "; " + t.get("note", "")}
    except (OSError, ValueError, KeyError):
        return {"traffic": None, "traffic_note": "PMC counters cannot be read inside this run; see profiles/pmc_traffic.json"}


def opening_measure(ctx, n=131072, reps=3):
    """Whisk tracker-opening proofs verified/s (IsValidWhiskOpeningProof, whisk_interface.py:147-169): n proofs handed over as
    packed wire bytes, verdicts back -- the device front-end (cg1_opening_prepare_device) and the host one.  The reference's valid goldens, cycled."""
    from curdleproofs_pie_amd.shuffle_verifier import OpeningBatchVerifier

    with open(os.path.join(ROOT, "tests", "golden", "opening_vectors.json")) as f:
        cases = json.load(f)["cases"]
    one = [(bytes.fromhex(c["r_G"]) + bytes.fromhex(c["k_r_G"]), bytes.fromhex(c["k_commitment"]), bytes.fromhex(c["proof"])) for c in cases]
    reps_n = (n + len(one) - 1) // len(one)
    trk = (b"".join(t for t, _, _ in one) * reps_n)[: 96 * n]
    kcs = (b"".join(k for _, k, _ in one) * reps_n)[: 48 * n]
    pfs = (b"".join(p for _, _, p in one) * reps_n)[: 128 * n]
    out = {"unit": "opening proofs/s", "n": n, "data": "the reference's %d valid golden proofs cycled; wire bytes in host memory -> verdicts" % len(one)}
    for key, dev in (("device_front_end", True), ("host_front_end", False)):
        v = OpeningBatchVerifier(ctx, device_front_end=dev)
        assert all(v.verify_packed(trk, kcs, pfs))
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            ok = v.verify_packed(trk, kcs, pfs)
            best = min(best, time.perf_counter() - t0)
            assert all(ok)
        out[key] = {"value": n / best, "ms_per_batch": best * 1e3}
    return out


def open_comm(args, rank, world, ctx):
    """The rank's communicator (None at N = 1): TCP control channel + RCCL for the data exchange unless the ranks share one GPU.
    If RCCL cannot be attached on some rank (library missing, ncclCommInitRank error) EVERY rank keeps the socket transport and the
    line says so in `collective.backend_note` -- the exchange is 144 bytes per rank either way."""
    if world == 1:
        return None
    from curdleproofs_pie_amd import _native as N
    from curdleproofs_pie_amd.distributed import init_comm

    t0 = time.perf_counter()
    comm = init_comm(rank, world, timeout_s=900.0)
    comm.note = None
    comm.rendezvous_s, comm.attach_s = time.perf_counter() - t0, None
    if args.backend == "rccl":
        try:
            t1 = time.perf_counter()
            comm.attach_rccl(ctx)                               # ncclGetUniqueId + ncclCommInitRank: collective, seconds on a cold node
            comm.attach_s = time.perf_counter() - t1
            err = b""
        except N.NativeError as e:
            err = str(e).encode()[:200]
        errs = [x.rstrip(b"\0") for x in comm.allgather(err.ljust(200, b"\0"), host_only=True)]
        if any(errs):
            bad = [(r, x.decode(errors="replace")) for r, x in enumerate(errs) if x]
            if not args.same_device and not args.allow_socket:
                # every rank has its own GPU and was asked for RCCL: north_star's exchange is RCCL over xGMI -- a line measured over the
                # TCP star would not be that measurement.  Every rank sees the same `errs`, so every rank leaves here.
                comm.barrier()
                comm.close()
                sys.exit("bench.py: RCCL could not be attached (rank %d: %s); re-run with --allow-socket to measure over the TCP transport instead" % bad[0])
            comm.barrier()
            comm.close()                                       # some ranks may hold a half-made RCCL communicator: every rank starts over
            comm = init_comm(rank, world, timeout_s=900.0)     # (rank 0 removed the rendezvous file after the first connect and writes it anew)
            comm.rendezvous_s, comm.attach_s = time.perf_counter() - t0, None
            comm.note = "RCCL not attached (rank %d: %s): socket transport" % bad[0]
            if rank == 0:
                print("bench.py: " + comm.note, file=sys.stderr, flush=True)
    return comm


def verify_mode(args, rank, local_rank, world):
    """BASELINE config 3 (N = 1) / config 5's structure (N > 1, proof-per-GPU): `--batch` proofs per GPU per step, from wire
    bytes in host memory to verdicts.  Ranks share the host's cores evenly.  No data-path collective."""
    from curdleproofs_pie_amd import _native as N
    from curdleproofs_pie_amd.distributed import max_over_ranks

    dev_index = 0 if args.same_device else local_rank
    ctx = N.Context(dev_index)
    comm = open_comm(args, rank, world, ctx)
    cores = int(N.cg1_shuffle_default_threads())         # usable CPUs (affinity mask capped by the cgroup quota)
    threads = max(1, cores // world)
    if comm:
        comm.barrier()
    out = verify_measure(ctx, threads, args.steps, args.warmup, args.batch, args.verify_mode, cpu_leg=(rank == 0 and not args.no_cpu_baseline),
                         front_end=args.front_end)
    el = out["ms_per_step"] * args.steps / 1e3
    per_rank = max_over_ranks(el, comm)
    # a core-starved run must be visible: every rank's front-end time and thread count travel to rank 0
    fe = max_over_ranks(out["phases_ms_per_step"][FE_KEY], comm)
    el = max(per_rank)
    if rank == 0:
        if world == 1:
            out["opening_proofs"] = opening_measure(ctx)
            out["single_proof"] = single_proof_measure(ctx)
        out.update({"value": world * args.batch * args.steps / el, "ms_per_step": el / args.steps * 1e3, "n_gpus": world, "scaling": "weak",
                    "vs_baseline": None, "dtype": "u32",
                    "per_rank": {"ms_per_step": [x / args.steps * 1e3 for x in per_rank], "front_end_ms_per_step": fe,
                                 "host_threads": threads, "host_cores_usable": cores,
                                 "note": "proofs/s is host-bound when front_end_ms_per_step approaches ms_per_step: each rank runs the "
                                         "verifier front-end on cores // n_gpus threads"},
                    "config": {"workload": "whisk_shuffle_verify ell=124 batch=%d per GPU" % args.batch,
                               "parallelism": "proof-per-GPU x%d, no data-path collective" % world}})
        if comm:
            out["collective"] = {"backend": comm.transport, "world_seen": comm.world_seen, "what": "none on the data path; clocks over the control channel"}
        print(json.dumps(out), flush=True)
    if comm:
        comm.barrier()
        comm.close()


def side_mode(args):
    """Secondary measurements (single GPU, not the driver's contract line)."""
    from curdleproofs_pie_amd import _native as N

    ctx = N.Context(0)
    d_g = ctx.alloc(96)
    d_g.upload(raw96_gen())
    if args.mode == "batched":
        M, n = 1024, 627                       # 5*ell + 7 at ell = 124 (Whisk N = 128), SURVEY 3.2
        tot = M * n
        d_k, d_pts, d_sc = ctx.alloc(32 * tot), ctx.alloc(96 * tot), ctx.alloc(32 * tot)
        ctx.gen_scalars_device(d_k, tot, 1); ctx.batch_mul_device(d_g, 1, d_k, d_pts, tot); ctx.gen_scalars_device(d_sc, tot, 2)
        offs = [n * j for j in range(M + 1)]
        for _ in range(args.warmup):
            ctx.msm_batched_device(d_pts, d_sc, offs)
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.msm_batched_device(d_pts, d_sc, offs)
        ctx.sync(); el = time.perf_counter() - t0
        print(json.dumps({"metric": "final-accumulator MSMs/sec (1024 independent 627-term MSMs per step; the MSM content of "
                                    "BASELINE config 3, not whole-proof verification)", "value": M * args.steps / el,
                          "unit": "MSMs/s", "scalar_muls_per_s": tot * args.steps / el, "n_gpus": 1, "steps": args.steps,
                          "warmup": args.warmup, "ms_per_step": el / args.steps * 1e3, "higher_is_better": True,
                          "phases_ms": ctx.timings()}))
    else:
        n = 1 << args.logn
        d_k, d_pts, d_sc = ctx.alloc(32 * n), ctx.alloc(96 * n), ctx.alloc(32 * n)
        ctx.gen_scalars_device(d_k, n, 1); ctx.batch_mul_device(d_g, 1, d_k, d_pts, n); ctx.gen_scalars_device(d_sc, n, 2)
        hp, hs = d_pts.download(), d_sc.download()
        for _ in range(args.warmup):
            ctx.msm_host(hp, hs, n)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            ctx.msm_host(hp, hs, n)
        el = time.perf_counter() - t0
        print(json.dumps({"metric": "BLS12-381 G1 scalar-muls/sec at MSM size 2^%d, inputs in pageable HOST memory (PCIe-inclusive; "
                                    "never the headline value)" % args.logn, "value": n * args.steps / el, "unit": "G1 scalar-muls/s",
                          "ms_per_step": el / args.steps * 1e3, "steps": args.steps, "warmup": args.warmup}))


# ---------------------------------------------------------------------------------------------- the MSM headline
class Workload:
    """Synthetic inputs of one rank, generated on the GPU and left resident in HBM: P_i = k_i * G (get_random_point,
    util.py:67-68), scalars uniform in [1, r-1] (util.py:21-24), both splitmix64-seeded."""

    def __init__(self, ctx, d_g, n_local, seed, groups=(0,)):
        self.n = n_local * len(groups)
        self.d_k = ctx.alloc(32 * self.n)
        self.d_pts, self.d_sc = ctx.alloc(96 * self.n), ctx.alloc(32 * self.n)
        for j, g in enumerate(groups):          # point group g of the job; a window-only rank holds every group
            ctx.gen_scalars_device(self.d_k.ptr + 32 * n_local * j, n_local, 0xC0FFEE + 7919 * seed + 1000 * g)
            ctx.batch_mul_device(d_g, 1, self.d_k.ptr + 32 * n_local * j, self.d_pts.ptr + 96 * n_local * j, n_local)
            ctx.gen_scalars_device(self.d_sc.ptr + 32 * n_local * j, n_local, 0xBEEF + 7919 * seed + 1000 * g)

    def closed_form_scalar(self):
        """sum_i k_i * s_i mod r over this rank's terms: the points are k_i * G, so the MSM of these terms is exactly that multiple of G"""
        kb, sb = memoryview(self.d_k.download()), memoryview(self.d_sc.download())
        f = int.from_bytes
        return sum(f(kb[o: o + 32], "little") * f(sb[o: o + 32], "little") for o in range(0, 32 * self.n, 32)) % FR_ORDER

    def free(self):
        self.d_pts.free(); self.d_sc.free(); self.d_k.free()


def launch_ranks(args):
    """`python bench.py --gpus N` with no launcher around it: this process touches no GPU -- it builds the library (hipcc, no HIP
    call), then starts the N ranks as fresh child processes (one per GPU, LOCAL_RANK = device ordinal), relays rank 0's JSON line
    and exits with the worst child status.  The ranks find each other through a rendezvous file (distributed.init_comm)."""
    import shutil
    import subprocess
    import tempfile

    from curdleproofs_pie_amd import build as B
    B.build(verbose=False)
    n = args.gpus
    tmp = tempfile.mkdtemp(prefix="cg1_bench_rdzv_")
    env = dict(os.environ, WORLD_SIZE=str(n), CG1_RDZV_FILE=os.path.join(tmp, "rdzv"), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    procs = []
    try:
        for r in range(n):
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                                          stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=(r == 0)))
        import threading

        lines = []
        reader = threading.Thread(target=lambda: lines.extend(procs[0].stdout), daemon=True)
        reader.start()
        rcs = [None] * n
        while any(rc is None for rc in rcs):
            for i, pr in enumerate(procs):
                if rcs[i] is None:
                    rcs[i] = pr.poll()
            if any(rc not in (None, 0) for rc in rcs):          # one rank failed: the others would wait for it until their timeout
                for i, pr in enumerate(procs):
                    if rcs[i] is None:
                        pr.terminate()
                for i, pr in enumerate(procs):
                    if rcs[i] is None:
                        try:
                            rcs[i] = pr.wait(timeout=20)
                        except subprocess.TimeoutExpired:
                            pr.kill()
                            rcs[i] = pr.wait()
                break
            time.sleep(0.05)
        reader.join(timeout=10)
        sys.stdout.write("".join(lines))
        sys.stdout.flush()
        bad = [rc for rc in rcs if rc]
        if bad:
            sys.exit("bench.py: rank exit codes %s" % rcs)
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        shutil.rmtree(tmp, ignore_errors=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--logn", type=int, default=20, help="log2 of terms per GPU")
    ap.add_argument("--window", type=int, default=0, help="0 = the library's plan for the size (c = 16 at 2^20); 4..16 uniform; -16..-4 balanced")
    ap.add_argument("--pipeline", type=int, default=1, choices=[1, 2],
                    help="MSM calls in flight per GPU in the timed region: 1 (default) = strictly one call after the other, so ms_per_step is "
                         "the latency of one call and the kernel times are those of kernels that own the chip; 2 = two contexts take the steps "
                         "alternately (cg1_msm_device_begin / _end): a step's sort phases run under the previous step's bucket accumulation and "
                         "its reduction tail under the next one's (always reported beside the headline as `two_calls_in_flight`)")
    ap.add_argument("--seed", type=int, default=1, help="input seed of the timed region (seeds 1, 2, 3 are also reported side by side)")
    ap.add_argument("--shard", choices=["hybrid", "windows", "points"], default="hybrid",
                    help="N>1: hybrid = 2 window-bucket groups x N/2 point groups (default); windows / points = pure splits")
    ap.add_argument("--cpu-sample-logn", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the proofs-verified/s object and the extra N>1 records")
    ap.add_argument("--no-python-face", action="store_true", help="skip the python_face / unchanged_control_flow objects (compute_MSM over G1Point / Scalar "
                                                                  "objects at 2^16 and 2^20; the reference's own call sequence replayed)")
    ap.add_argument("--batch", type=int, default=None, help="proofs per GPU per step (secondary metric / --mode verify): 1024 by default -- BASELINE "
                                                            "config 3 -- and 2048 for --mode verify at N > 1 (config 5's 16 384 proofs over 8 GPUs)")
    ap.add_argument("--verify-steps", type=int, default=40, help="batches of the secondary stream (its fill and drain are inside the timed region)")
    ap.add_argument("--verify-mode", choices=["merged", "independent"], default="merged")
    ap.add_argument("--front-end", choices=["auto", "host", "device"], default="auto",
                    help="verifier front-end (transcript, D / A', challenge algebra): host threads, k_shuffle_front_end, or by the threads available")
    ap.add_argument("--mode", choices=["msm", "batched", "pcie", "verify"], default="msm",
                    help="msm: the driver's line (headline MSM + secondary proofs/s). batched: BASELINE config 3's MSM content (1024 independent "
                         "627-term accumulator MSMs per step, regime B). verify: BASELINE config 3 end to end alone (and config 5's structure at "
                         "N>1). pcie: the headline MSM with inputs in HOST memory")
    ap.add_argument("--backend", choices=["rccl", "socket", "nccl", "gloo"], default="rccl",
                    help="transport of the N>1 partial-sum exchange: rccl (ncclAllGather over xGMI, one GPU per rank) or socket (the TCP "
                         "control channel alone: ranks rehearsing on one GPU); nccl / gloo are accepted as aliases")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses GPU 0 (implies --backend socket)")
    ap.add_argument("--allow-socket", action="store_true", help="N > 1 with one GPU per rank: keep going on the TCP transport when RCCL cannot be attached "
                                                                 "(default: exit non-zero -- a scaling run must not silently stop being an RCCL run)")
    args = ap.parse_args()

    if args.batch is None:
        args.batch = 2048 if (args.mode == "verify" and args.gpus > 1) else 1024
    if args.backend in ("nccl", "gloo"):                      # the names torch gives the same two transports
        args.backend = {"nccl": "rccl", "gloo": "socket"}[args.backend]
    if args.same_device:
        os.environ["CURDLE_G1_SHARE_DEVICE"] = "1"            # ranks beyond the device count are meant to share GPU 0 here (N.default_context() refuses otherwise)
    if args.same_device and not os.environ.get("CG1_BENCH_TRY_RCCL_ON_ONE_DEVICE"):
        args.backend = "socket"                               # RCCL refuses two ranks on one device (the env switch lets a test see that refusal handled)
    if args.gpus > 1 and "RANK" not in os.environ:
        return launch_ranks(args)                             # no launcher around us: spawn the ranks ourselves
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        args.gpus = world

    # build BEFORE the library is first loaded: a stale .so must not stay mapped under a fresh one
    from curdleproofs_pie_amd import build as B
    B.build(verbose=False)
    from curdleproofs_pie_amd import _native as N

    if args.mode == "verify":
        return verify_mode(args, rank, local_rank, world)
    if args.mode != "msm":
        return side_mode(args)
    dev_index = 0 if args.same_device else local_rank
    ctx = N.Context(dev_index)
    ctx2 = N.Context(dev_index)                               # second pipeline of the same GPU (own stream and scratch buffers)
    comm = open_comm(args, rank, world, ctx)

    from curdleproofs_pie_amd.distributed import all_reduce_g1, max_over_ranks, shard_layout

    c = args.window
    d_g = ctx.alloc(96)
    d_g.upload(raw96_gen())

    def barrier_sync():
        ctx.sync()                                            # hipDeviceSynchronize: this rank's GPU is idle ...
        if comm:
            comm.barrier()                                    # ... and so is every other rank's

    def run(shard, n_per_gpu, seed, steps, warmup, depth=None, detail=False, check=False):
        """One timed measurement: ONE MSM of world * n_per_gpu terms per step, sharded `shard`-wise, `depth` calls in flight.
        Returns a dict with the max-over-ranks time and what the audit needs."""
        depth = depth or args.pipeline
        ctxs = [ctx, ctx2][:depth]
        for cx in ctxs:                                       # 1: hipEvents around k_accumulate only (the roofline kernel); 2: around every
            cx.set_param("profile", 2 if detail else 1)       # phase -- 8 marker packets of ~5.5 us each, kept out of the timed region
        w_rank, w_groups, p_rank, p_groups = shard_layout(rank, world, shard)
        n_group = n_per_gpu * world // p_groups            # terms of one point group (= of this rank)
        wl = Workload(ctx, d_g, n_per_gpu, seed, groups=[p_rank * (world // p_groups) + j for j in range(world // p_groups)])
        assert wl.n == n_group
        ex = {"t": 0.0, "n": 0}

        phase_acc = {}

        def exchange(part):
            if world == 1:
                return part
            t = time.perf_counter()
            out = all_reduce_g1(part, comm)
            ex["t"] += time.perf_counter() - t
            ex["n"] += 1
            return out

        def finish(cx):
            part = cx.msm_device_end()
            for k, v in cx.timings().items():
                phase_acc[k] = phase_acc.get(k, 0.0) + v
            return exchange(part)

        def stream(count):
            """`count` complete steps; with depth 2 step k+1 is enqueued on the other context before step k is collected"""
            out, pending = [], None
            for k in range(count):
                cx = ctxs[k % len(ctxs)]
                cx.msm_device_begin(wl.d_pts, wl.d_sc, wl.n, window_c=c, shard_rank=w_rank, shard_world=w_groups)
                if len(ctxs) == 1:
                    out.append(finish(cx))
                    continue
                if pending is not None:
                    out.append(finish(pending))
                pending = cx
            if pending is not None:
                out.append(finish(pending))
            return out

        # bring the chip to the clock it holds under load before anything is timed: after the idle spell of input generation the first
        # ~50 ms of work run at a ramping clock (the same kernels measure 28 T mad/s there and 33 once warm: profiles/r03_v2_bench.json)
        ctx.probe_mad_rate(2, 200)
        results = stream(warmup)
        # ... and let the step time settle (untimed, reported as settle_steps): blocks of five further steps until a block is within 1.5 %
        # of the one before -- the first steps of a process also pay for host threads and pages that are touched for the first time
        # (profiles/r03_v4_bench.json: 3.24 ms for the first 20 timed steps of a process, 2.98 / 3.00 ms for the next two runs)
        # (two agreeing comparisons in a row, i.e. three blocks at one level: a single pair has been seen to agree on the way down --
        # one round-4 run timed 3.14 ms after 20 settling steps and 2.98 / 3.00 ms in its next two runs)
        settle, prev, calm = 0, None, 0
        while settle < 60:
            ts = time.perf_counter()
            results += stream(5)
            settle += 5
            blk = (time.perf_counter() - ts) / 5
            calm = calm + 1 if (prev is not None and abs(blk - prev) <= 0.015 * prev) else 0
            done = calm >= 2
            prev = blk
            if comm:                                          # every rank takes the same number of steps: stop when all have settled
                done = max(max_over_ranks(0.0 if done else 1.0, comm)) < 0.5
            if done:
                break
        ex["t"], ex["n"] = 0.0, 0
        phase_acc.clear()
        import gc
        gc.collect()
        gc.disable()                                        # (the interpreter's cyclic collector must not land inside the 20 timed steps: it does none of the step's work)
        barrier_sync()
        t0 = time.perf_counter()
        results += stream(steps)                            # every step is begun AND collected inside the timed region
        barrier_sync()
        elapsed = mine = time.perf_counter() - t0
        gc.enable()
        counts = ctx.last_counts()
        if not all(N.cg1_eq(r, results[0]) for r in results[1:]):
            sys.exit("bench.py: MSM results differ between steps")
        closed = None
        if check:
            # the timed MSM against its closed form: every point group's sum_i k_i s_i (taken from the ranks of window group 0 --
            # the other window groups hold the same points), added up, times G by the host operator (double-and-add, host_g1.cpp)
            import ctypes
            mine_sum = wl.closed_form_scalar() if w_rank == 0 else 0
            parts = comm.allgather(mine_sum.to_bytes(32, "little"), host_only=True) if comm else [mine_sum.to_bytes(32, "little")]
            tot = sum(int.from_bytes(b, "little") for b in parts) % FR_ORDER
            g, want = ctypes.create_string_buffer(N.POINT_BYTES), ctypes.create_string_buffer(N.POINT_BYTES)
            N.cg1_generator(g)
            N.cg1_mul(want, g.raw, tot.to_bytes(32, "little"))
            closed = bool(N.cg1_eq(results[-1], want.raw))
            if not closed:
                sys.exit("bench.py: the timed MSM's result differs from its closed form (sum k_i s_i) * G")
        per_rank = max_over_ranks(mine, comm)
        elapsed = max(per_rank)
        rec = {"elapsed": elapsed, "ms_per_step": elapsed / steps * 1e3, "value": n_per_gpu * world * steps / elapsed,
               "per_rank_ms_per_step": {"min": min(per_rank) / steps * 1e3, "max": max(per_rank) / steps * 1e3},
               "phases_ms": {k: v / steps for k, v in phase_acc.items() if k != "window_c"}, "counts": counts,
               "phases_ms_window_c": phase_acc.get("window_c", 0) / steps,
               "layout": (w_rank, w_groups, p_rank, p_groups), "n_local": wl.n, "depth": len(ctxs),
               "exchange_ms": (ex["t"] / ex["n"] * 1e3) if ex["n"] else None, "result": results[-1], "closed_form_ok": closed, "settle_steps": settle}
        wl.free()
        return rec

    n_per_gpu = 1 << args.logn
    main_rec = run(args.shard, n_per_gpu, args.seed, args.steps, args.warmup, check=True)
    # the per-phase table: a few more steps with an event around every phase (not the timed region)
    main_rec["phases_ms_detailed"] = run(args.shard, n_per_gpu, args.seed, max(3, args.steps // 4), 1, detail=True)["phases_ms"]

    extra = {}
    if not args.no_secondary:
        # the reference's input distribution at three seeds: short side-by-side runs outside the timed region
        seeds = {}
        for s in (1, 2, 3):
            seeds[str(s)] = main_rec["ms_per_step"] if s == args.seed else run(args.shard, n_per_gpu, s, max(3, args.steps // 2), 1)["ms_per_step"]
        extra["seeds_ms_per_step"] = dict(seeds, min=min(seeds.values()), median=statistics.median(seeds.values()))
        other = run(args.shard, n_per_gpu, args.seed, args.steps, args.warmup, depth=3 - main_rec["depth"])
        extra["two_calls_in_flight" if main_rec["depth"] == 1 else "single_call"] = {
            "what": ("the same steps with TWO MSM calls in flight on the GPU (two contexts, cg1_msm_device_begin / _end): throughput of a stream of "
                     "MSMs; kernel times stretch because kernels of both calls share the chip" if main_rec["depth"] == 1 else
                     "the same steps strictly one after the other (one MSM call in flight): per-call latency"),
            "ms_per_step": other["ms_per_step"], "value": other["value"], "phases_ms": other["phases_ms"]}
        if world > 1:
            # what north_star names literally: window buckets sharded over ALL N GPUs, every rank holding all N * 2^20 points
            if args.shard != "windows":
                w = run("windows", n_per_gpu, args.seed, args.steps, args.warmup)
                extra["windows_only"] = {"ms_per_step": w["ms_per_step"], "value": w["value"], "per_rank_ms_per_step": w["per_rank_ms_per_step"],
                                         "input_bytes_resident_per_rank": 128 * n_per_gpu * world,      # every rank holds ALL points and scalars
                                         "same_result_as_default_shard": bool(N.cg1_eq(w["result"], main_rec["result"])),
                                         "predicted_ms_per_step_emulated": EMULATED_MS["windows"].get(world)}
            # BASELINE config 4: ONE MSM of 2^22 terms in total over the N GPUs (strong scaling)
            if (1 << 22) % world == 0:
                sc = run(args.shard, (1 << 22) // world, args.seed, args.steps, args.warmup)
                extra["strong_2_22_total"] = {"terms_total": 1 << 22, "ms_per_step": sc["ms_per_step"], "value": sc["value"],
                                              "per_rank_ms_per_step": sc["per_rank_ms_per_step"], "scaling": "strong"}

    if rank == 0:
        r = main_rec
        w_rank, w_groups, p_rank, p_groups = r["layout"]
        acc_ms = r["phases_ms"]["accumulate"]                 # per step: BOTH k_accumulate launches when the call ran as two chains
        launches = max(1, r["counts"].get("accumulate_launches", 1))
        # roofline of the dominant kernel (k_accumulate): ALGORITHMIC bytes = 128 B per (point, scalar) term
        # (96 B affine point + 32 B scalar) x the terms one launch processes
        achieved = 128.0 * r["n_local"] / (acc_ms * 1e-3) / 1e9
        # additions k_accumulate really performs: a chunk's first entry is a copy and its first addition is the cheaper
        # affine + affine form.  Every chunk is charged one of those (a one-entry chunk performs none, so this undercounts).
        madds = r["counts"]["mixed_adds"]                   # bucket entries - chunks
        pair_adds = min(r["counts"]["chunks"], madds)
        mads = (madds - pair_adds) * float(MADS_PER_MADD) + pair_adds * float(MADS_PER_MMADD)
        pmc = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and world == 1 and args.logn == 20:
            try:
                pmc = json.load(open(tpath))
            except Exception:
                pmc = None
        ctx.set_param("profile", 0)
        peak = mad_peak_same_run(ctx)
        out = {
            "metric": "BLS12-381 G1 scalar-muls/sec at MSM size 2^20",
            "value": r["value"],
            "unit": "G1 scalar-muls/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "settle_steps": r["settle_steps"],              # further untimed steps, taken until the step time stops moving (see run())
            "ms_per_step": r["ms_per_step"],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u32",
            "data": "synthetic: points k_i*G and scalars uniform in [1,r-1], splitmix64-seeded (seed %d), generated on the GPU" % args.seed,
            "config": {"workload": f"single MSM of 2^{args.logn} x {world} BLS12-381 G1 terms, resident in HBM, "
                                   f"sharded over {world} GPU(s): {w_groups} window-bucket group(s) x {p_groups} point group(s)",
                       "terms_total": n_per_gpu * world, "terms_per_gpu": n_per_gpu, "window_c": int(r["phases_ms_window_c"]), "shard": args.shard,
                       "parallelism": f"windows x{w_groups} . points x{p_groups}, one all-gather of {world} partial G1 sums",
                       "calls_in_flight": r["depth"],
                       "arithmetic": "381-bit Fp as 14 x 28-bit limbs in u32, Montgomery, 64-bit column accumulators (v_mad_u64_u32)",
                       "device_build": N.build_info().get("pipeline", "unknown"),
                       "result_equals_closed_form": r["closed_form_ok"],
                       "bit_exact_vs_oracle": "tests/test_msm_gpu.py"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, **pmc_traffic(args.logn if world == 1 else 0),
                         "kernel": "k_accumulate", "kernel_ms": acc_ms / launches, "launches_per_step": launches, "kernel_ms_per_step": acc_ms,
                         "launch_note": ("a call of this size runs as TWO launch chains (high / low half of the windows, two streams): each "
                                         "k_accumulate launch takes every term through half of the windows, i.e. 64 B of the term's 128 algorithmic "
                                         "bytes; achieved = 128 B x terms / the two launches' summed duration = 64 B x terms / one launch's average"
                                         if launches == 2 else "one k_accumulate launch per call"),
                         "note": "path is integer-multiply (v_mad_u64_u32) bound, not HBM-bound: see roofline_int_mad and DESIGN.md"},
            # The bound that actually applies: 32x32+64 integer multiply-adds.  Algorithmic MADs of one
            # k_accumulate launch = the additions it really performs (bucket entries minus one copy per chunk, both counted on
            # the device in this run): 3542 each, 1974 for the first one of a chunk; peak = the chip-wide v_mad_u64_u32 rate
            # measured by tools/ubench_valu.hip.
            "roofline_int_mad": {"bound": "valu v_mad_u64_u32", "achieved": mads / (acc_ms * 1e-3) / 1e12, "peak": peak["peak_T"],
                                 "unit": "T mad/s", "frac": mads / (acc_ms * 1e-3) / (peak["peak_T"] * 1e12), "kernel": "k_accumulate",
                                 "kernel_ms": acc_ms / launches, "launches_per_step": launches, "kernel_ms_per_step": acc_ms, "peak_same_run": peak,
                                 "peak_round1_other_box": MAD_PEAK_T,
                                 "mixed_adds_per_launch": madds, "bucket_entries": r["counts"]["entries"], "chunks": r["counts"]["chunks"],
                                 "mads_per_mixed_add": MADS_PER_MADD, "first_additions_of_chunks": pair_adds, "mads_per_first_addition": MADS_PER_MMADD,
                                 "valu_per_wave_mixed_add": (pmc or {}).get("k_accumulate_valu_per_wave_mixed_add"),
                                 "valu_source": (pmc or {}).get("sq_source")},
            "phases_ms": dict(r["phases_ms_detailed"], note="from extra steps with a hipEvent around every phase; the timed region brackets only "
                                                                 "k_accumulate (roofline.kernel_ms)"),
            "per_rank_ms_per_step": r["per_rank_ms_per_step"],
        }
        out.update(extra)
        if world > 1:
            out["collective"] = {"backend": comm.transport, "backend_note": getattr(comm, "note", None), "world_seen": comm.world_seen,
                                 "world_seen_source": "ncclCommCount" if comm.transport == "rccl" else "ranks connected to the TCP hub",
                                 "what": "cg1_comm_allreduce_g1: all-gather of one 144-byte partial G1 sum per rank (ncclAllGather on the context's "
                                         "stream when the backend is rccl), then world-1 host additions on every rank",
                                 "bytes_per_step": 144 * world, "ms_per_exchange": r["exchange_ms"],
                                 "rendezvous_s": getattr(comm, "rendezvous_s", None), "rccl_attach_s": getattr(comm, "attach_s", None)}
            out["measured_on"] = ("%d ranks, one GPU each: a measured multi-GPU line" % world) if not args.same_device else \
                                 ("%d ranks REHEARSING on one GPU (--same-device): not a scaling measurement" % world)
            out["predicted_ms_per_step_emulated"] = EMULATED_MS.get(args.shard, {}).get(world)
        if world == 1:
            # the practical ceiling of the same arithmetic: chains of dependent mixed additions on register-resident operands
            # (k_probe_madd: no sort, no gathers, every lane busy), as many lane-level additions as one k_accumulate launch
            wl = Workload(ctx, d_g, 4096, args.seed)
            lanes, iters = 1 << 19, 32
            ctx.set_param("profile", 0)
            pms = min(ctx.probe_madd(wl.d_pts, 4096, lanes, iters) for _ in range(3))
            wl.free()
            probe_rate = lanes * iters / (pms * 1e-3)
            acc_rate = mads / float(MADS_PER_MADD) / (acc_ms * 1e-3)       # k_accumulate in mixed-add equivalents per second
            out["roofline_int_mad"]["madd_probe"] = {
                "what": "k_probe_madd: %d lanes x %d dependent mixed additions, operands in registers" % (lanes, iters),
                "ms": pms, "lane_mixed_adds_per_s": probe_rate, "k_accumulate_mixed_add_equivalents_per_s": acc_rate,
                "k_accumulate_vs_probe": acc_rate / probe_rate}
        if world == 1 and not args.no_cpu_baseline:
            wl = Workload(ctx, d_g, max(1 << 18, 1 << args.cpu_sample_logn), args.seed)
            out["cpu_baseline"] = cpu_baseline(wl.d_pts, wl.d_sc, 1 << args.cpu_sample_logn)
            wl.free()
        if world == 1 and not args.no_secondary:
            ctx2.close()                                      # (its streams would keep hardware queues the verifier's lanes need: a process has 24)
            cores = int(N.cg1_shuffle_default_threads())
            out["secondary"] = verify_measure(ctx, cores, args.verify_steps, 2, args.batch, args.verify_mode, cpu_leg=not args.no_cpu_baseline, peak_T=peak["peak_T"], front_end=args.front_end)
            keys = ("value", "unit", "ms_per_step", "steps", "batch", "front_end", "host_threads", "phases_ms_per_step")
            if out["secondary"]["front_end"].startswith("device"):
                # the same stream with the front-end on the host's threads (what a verifier with >= 24 threads to itself would run)
                hv = verify_measure(ctx, cores, args.verify_steps, 1, args.batch, args.verify_mode, cpu_leg=False, peak_T=peak["peak_T"], front_end="host")
                out["secondary"]["host_front_end_on_all_host_threads"] = {k: hv[k] for k in keys}
            # the device front-end with TWO host threads: what a rank gets on a host shared by many GPUs
            dv = verify_measure(ctx, 2, args.verify_steps, 1, args.batch, args.verify_mode, cpu_leg=False, peak_T=peak["peak_T"], front_end="device")
            out["secondary"]["device_front_end_on_2_host_threads"] = {k: dv[k] for k in keys}
            # ... and with 2048 proofs per batch: BASELINE config 5's share of one GPU (16 384 proofs over 8)
            dv2 = verify_measure(ctx, 2, max(10, args.verify_steps // 2), 1, 2 * args.batch, args.verify_mode, cpu_leg=False, peak_T=peak["peak_T"], front_end="device")
            out["secondary"]["device_front_end_on_2_host_threads_batches_of_%d" % (2 * args.batch)] = {k: dv2[k] for k in keys if k != "phases_ms_per_step"}
            out["secondary"]["opening_proofs"] = opening_measure(ctx)
            out["secondary"]["single_proof"] = single_proof_measure(ctx)
        if world == 1 and not args.no_python_face:
            # the drop-in itself: compute_MSM / MSMAccumulator through the reference's signature (lists of G1Point / Scalar objects), and
            # the reference's own backend-call sequence for one proof replayed through this backend.  (After the secondary metric: these
            # run on the package's default context, and every further live context costs the verifier hardware queues -- DESIGN.md 9.)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import gpu_python_face
            import replay_call_trace
            out["python_face"] = gpu_python_face.measure(sizes=(16, 20), reps=3)
            out["unchanged_control_flow"] = replay_call_trace.measure(reps=3)
            import curdleproofs_pie_amd.msm_accumulator as MA
            MA.release()
            N.close_default_context()
        print(json.dumps(out), flush=True)

    if comm:
        comm.barrier()
        comm.close()


if __name__ == "__main__":
    main()
