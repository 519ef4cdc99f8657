#!/usr/bin/env python3
"""Headline benchmark: CG iterations/sec + SpMV achieved HBM GB/s, 256^3 Poisson (BASELINE.json).

    python bench.py --gpus N --steps K --warmup W

A "step" is one CG iteration (one pass of the hot path: stencil SpMV + the BLAS-1 ops of
SolverCg.hpp:96-123) on the per-GPU 256^3 7-point Poisson block of BASELINE.json configs[1]
(synthetic structured-as-unstructured face graph, b = 1, x0 = 0, fp64, tolerances disabled so
exactly K iterations run).  For N > 1 (one rank per GPU) the global mesh is 256 x 256 x (256 N)
row-partitioned in z-slabs (weak scaling): halo planes and dot-product all-reduces move inside
libstorm_hip.so (RCCL, or the library's peer-window transport).

Prints ONE JSON line on rank 0.
  value             256^3-block CG iterations per second summed over the GPUs (at N = 1 plain CG it/s) on the record
                    format the operator qualifies for (the natural-order box: byte-indexed canonical rows);
  value_general     the same solve on fp64 records, the format ANY mesh gets (Triangle, graded, renumbered);
  roofline          the SpMV kernel of the headline run from HIP-event pairs around every launch on the library's
                    compute stream during a second, identical solve.  `achieved` / `frac` price the bytes the
                    operator's record format REALLY streams per launch (records + x + y: a physical HBM fraction
                    <= 1).  `traffic` = HBM bytes per launch by PMC (FETCH_SIZE doubled + WRITE_SIZE, separate
                    rocprofv3 passes of a child process started by THIS run); when that measurement is unavailable
                    `traffic` is null and `traffic_from_profile` quotes the committed profile with its file hash.
                    SURVEY.md 8d's fp64-weight + int32-column bytes over the same time are `effective_vs_8d_GBs`
                    (may exceed the peak: the byte-indexed formats do not move those bytes -- not a bandwidth);
  spmv              the SpMV ALONE as SURVEY.md 8d specifies it (the "SpMV achieved HBM GB/s" half of the metric): >= 50
                    stand-alone storm_hip_op_apply launches on x_i = sin(0.37 i) after warm-up, HIP-event median and mean,
                    for the lattice records (`spmv.lattice`) and the fp64 records any mesh gets (`spmv.general`), priced by
                    8d's bytes, by the streamed bytes and by PMC traffic; once back to back on one (x, y) pair and once
                    rotating three pairs (so the Infinity Cache's share shows);
  roofline_general  the same for the fp64-record kernel, where streamed bytes == 8d's algorithmic bytes;
  roofline_permuted_rcm  SURVEY.md 8d's unstructured stress variant: cells renumbered by the seeded permutation,
                    then the library's ordering from the cell centres (round 3: reverse Cuthill-McKee -- the key keeps its
                    name); whatever record format that operator gets;
  roofline_unstructured  the 256^3 graph with a jittered geometry (no two weights equal => fp64 records, SURVEY 8d's
                    bytes), cells renumbered by the seeded permutation, then the library's ordering: the number a
                    Triangle / TetGen mesh of this size would get;
  roofline_unstructured3d  a genuinely unstructured 3-D mesh at HBM scale: 12.6 M tetrahedra (seeded: six shapes of cells,
                    rows of 2 - 4 neighbours, all weights distinct) written as TetGen files, read back by the library's
                    reader (3-D branch of read_mesh_from_tetgen), Z-order numbering, fp64 records: SpMV fraction by SURVEY
                    8d's bytes (fused-dot launches of a CG solve AND the stand-alone `spmv` block), PMC traffic, ELL
                    padding, CSR tail, CG it/s;
  config1_cg64, config3_bicgstab256, config4_gmres30_convdiff128, config5_cavity128   BASELINE configs 1, 3, 4, 5 on this GPU, bounded;
  extra_gmres30_poisson256   GMRES(30) at the headline size (kernel-per-statement path: the Gram-Schmidt passes at HBM scale);
  host_loop_cg256   a USER's statement-by-statement CG (the reference's iterate() typed against Storm.hpp) with the library's lazy
                    statements / with every statement a launch / the library's device loop, us per iteration;
  multi_rank_path_at_one_rank   CG and BiCGStab on ONE rank over a size-1 RCCL communicator (a z-periodic box: both halo planes
                    exchanged with the rank itself, every reduction through the all-reduce) against the plain path, with the
                    device-timestamp comm_breakdown: what the N > 1 code path costs before any link latency;
  cpu_baseline      the CPU oracle (single thread, the reference is single-threaded) on a bounded sample.

N > 1: every rank process is a SUPERVISOR that never touches the GPU; it starts the measuring rank as a child with
a wall-clock budget.  The child runs a bounded pre-flight (one halo exchange + one all-reduce on a tiny slab stack,
checked numerically) before the 256^3 setup.  If any rank's child fails or exceeds the budget, all children are
killed and a FRESH set starts on the next transport of the chain.  The chain is rccl, ipc, host-staged: RCCL (the
transport north_star names) and the peer windows are BOTH measured, `value` is the better, `transports_measured` holds
both; host-staged runs only if neither worked.  The line records `transport`, `transports_measured`, `transport_fallback`.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--n", "--edge", dest="n", type=int, default=None,
                    help="cells per edge of the per-GPU block (use --edge under torch.distributed.run, whose own parser "
                         "claims the abbreviation --n); default 256 (--contact: 64)")
    ap.add_argument("--cpu-iters", type=int, default=20, help="CPU-baseline sample (CG iterations); 0 = skip")
    ap.add_argument("--cpu-parallel", action="store_true",
                    help="also time the OpenMP variant of the CPU baseline (3 runs; not the reference's loop order)")
    ap.add_argument("--ordering", default="natural", choices=["natural", "tile"])
    ap.add_argument("--nontemporal", type=int, default=-1)
    ap.add_argument("--opt", action="append", default=[], help="library option key=value (repeatable)")
    ap.add_argument("--spinup-seconds", type=float, default=1.5, help="untimed device spin-up before the warmup steps")
    ap.add_argument("--skip-general", action="store_true", help="skip the fp64-record repeat of the measurement")
    ap.add_argument("--skip-blas1", action="store_true", help="skip the per-kernel BLAS-1 rates")
    ap.add_argument("--shared-device", action="store_true",
                    help="debug: all ranks on device 0 (transport chain ipc,host); exercises the N > 1 code path of this "
                         "script on a one-GPU box -- the rates it prints mean nothing")
    ap.add_argument("--transport", default=None,
                    help="N > 1: comma-separated chain, each entry run by FRESH rank processes: rccl (halo send/recv + "
                         "all-reduce over RCCL: the transport BASELINE.json's north_star names), ipc (the library's peer-window "
                         "transport: hipIpc-mapped device memory, direct stores over xGMI, rank-ordered all-reduce fused into "
                         "the reductions' final pass), host (halo planes and scalars staged through host memory over gloo), "
                         "rccl-plain (rccl with cross-stream events, two-launch reductions, no early halo, no fused step: "
                         "run only if rccl gave no number).  Default rccl,rccl-plain,ipc,host (--contact: rccl,rccl-plain; "
                         "--shared-device: ipc,host): rccl AND ipc are both measured -- `value` is the better one, "
                         "`transports_measured` holds both with their comm_breakdown --, each guarded by the pre-flight before "
                         "and the post-flight check after its timed region; host only if neither worked")
    ap.add_argument("--one-transport", action="store_true", help="N > 1: stop at the first transport of the chain that works")
    ap.add_argument("--attempt-seconds", default="240,150,150",
                    help="wall-clock budget of the 1st, 2nd, 3rd (and every later) transport attempt (N > 1)")
    ap.add_argument("--inject-fail", default="", help="test hook: TRANSPORT=hang|exit|wrong|post[:RANK] makes that attempt fail "
                                                      "(a rank hangs / dies / the pre-flight / the post-flight finds wrong values)")
    ap.add_argument("--force-comm", action="store_true",
                    help="take the multi-rank code path (process group, RCCL communicator, all-reduces) even at N = 1")
    ap.add_argument("--min-seconds", type=float, default=0.25,
                    help="the K-step solve is repeated (each repeat bracketed by barrier + sync, max over ranks) until "
                         "the timed repeats add up to this; ms_per_step is the median repeat")
    ap.add_argument("--traffic", default="measure", choices=["measure", "profile", "off"],
                    help="roofline.traffic: measure = two rocprofv3 --pmc passes of a child process in this run (N = 1, "
                         "edge 256); profile = quote profiles/spmv_hbm_traffic.json as traffic_from_profile only")
    ap.add_argument("--skip-permuted", action="store_true", help="skip the permuted + RCM stress variant (SURVEY.md 8d)")
    ap.add_argument("--skip-unstructured", action="store_true", help="skip the jittered-geometry (fp64 records) variant")
    ap.add_argument("--skip-configs", action="store_true", help="skip BASELINE configs 3, 4, 5")
    ap.add_argument("--self-exchange", default="cg,bicgstab",
                    help="N = 1: solvers of the multi_rank_path_at_one_rank block (a size-1 RCCL communicator, child processes); '' = skip")
    ap.add_argument("--roofline-launches", type=int, default=200,
                    help="launches of the dominant kernel timed for `roofline` (at least this many, whatever --steps is)")
    ap.add_argument("--skip-unstructured3d", action="store_true", help="skip the tetrahedral mesh (roofline_unstructured3d)")
    ap.add_argument("--tet-edge", type=int, default=128, help="cubes per edge of the tetrahedral box (6 n^3 cells: 128 -> 12.6 M)")
    ap.add_argument("--tet-prefix", default=None, help=argparse.SUPPRESS)
    ap.add_argument("--skip-spmv", action="store_true", help="skip the stand-alone SpMV block (`spmv` in the line)")
    ap.add_argument("--spmv-launches", type=int, default=60, help="stand-alone SpMV launches timed per mode (SURVEY.md 8d: >= 50)")
    ap.add_argument("--spmv-only", action="store_true", help="run only the stand-alone SpMV block (for a clean rocprofv3 --stats comparison)")
    ap.add_argument("--spmv-what", default="lattice,general", help="--spmv-only: which operators (lattice, general, tets)")
    ap.add_argument("--pmc-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--detail-path", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the full record goes (everything the compact stdout line leaves out); '' = nowhere")
    ap.add_argument("--contact", action="store_true",
                    help="first-contact mode for a short multi-GPU lease (< 20 s of GPU work per transport): edge 64 unless "
                         "--edge is given, pre-flight, one RCCL CG and one RCCL BiCGStab with the device-timestamp breakdown, the "
                         "compact line; no stand-alone SpMV / stress / config blocks")
    args = ap.parse_args()
    if args.contact:
        contact_defaults(args)
    if args.n is None:
        args.n = 256
    chain = [t for t in (args.transport or default_chain(args)).split(",") if t]
    for t in chain:
        if t not in ("rccl", "rccl-plain", "ipc", "host"):
            ap.error(f"unknown transport {t!r}")

    if args.pmc_child:
        return pmc_child(args)
    if args.spmv_only:
        return spmv_only(args)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, args)
    if args.gpus > 1 and os.environ.get("STORM_BENCH_WORKER") != "1":
        return supervise(args, chain)
    transport = os.environ.get("STORM_BENCH_TRANSPORT", chain[0])
    if os.environ.get("STORM_BENCH_WORKER") == "1":
        # a rank that hangs inside a C call (RCCL bootstrap, a stream wait) never returns to the interpreter: the
        # default action of SIGALRM ends the process whatever it is doing; cancelled once the pre-flight has passed
        import signal

        signal.signal(signal.SIGALRM, signal.SIG_DFL)
        signal.alarm(int(os.environ.get("STORM_BENCH_PREFLIGHT_SECONDS", "120")))

    import numpy as np
    import torch

    from stormruler_amd import api, dist, mesh, partition

    rank, local_rank, world = dist.env_rank()
    if world != args.gpus:
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; launch with "
                  f"`python -m torch.distributed.run --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...`",
                  file=sys.stderr)
        return 2
    if not torch.cuda.is_available():
        print("bench.py needs an MI355X; the HIP path has no CPU fallback", file=sys.stderr)
        return 3
    if args.shared_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 or args.force_comm:
        os.environ.setdefault("RANK", "0"), os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"), os.environ.setdefault("MASTER_PORT", "29533")
        # torch.distributed only bootstraps (ids, window handles, barriers, the max of timings): gloo whenever the data
        # path itself is not RCCL, so that a rank pair RCCL cannot serve still has a control plane
        with stdout_to_stderr():  # (gloo announces its connections on stdout)
            dist.init_process_group("nccl" if (transport in ("rccl", "rccl-plain") and not args.shared_device) else "gloo")

    def connect(ctx_):
        if world == 1:
            if args.force_comm:
                ctx_.comm_init(api.Context.comm_unique_id(), 1, 0)
            else:
                dist.connect(ctx_)
        elif transport == "ipc":
            dist.connect_ipc(ctx_)
        elif transport == "host":
            dist.connect_host_staged(ctx_)
        else:
            dist.connect(ctx_)

    preflight, inject_post = None, False
    if world > 1:
        inject = dict(kv.split("=") for kv in args.inject_fail.split(",") if "=" in kv).get(transport, "")
        kind, _, who = inject.partition(":")
        hit = bool(kind) and (who == "" or int(who) == rank)
        if hit and kind == "exit":
            print(f"bench.py: injected failure on rank {rank} ({transport})", file=sys.stderr)
            os._exit(41)
        if hit and kind == "hang":
            import signal as _sig

            _sig.alarm(0)
            time.sleep(3600)
        inject_post = hit and kind == "post"
        preflight = run_preflight(api, dist, mesh, partition, connect, local_rank, world, rank, wrong=hit and kind == "wrong")
    if os.environ.get("STORM_BENCH_WORKER") == "1":
        import signal

        signal.alarm(0)

    n, K, W = args.n, args.steps, args.warmup
    t_setup = time.time()
    phases, t_phase = {}, [time.time()]  # wall seconds of this process by block (where a 100 s run spends its time)

    def phase(name):
        now = time.time()
        phases[name] = phases.get(name, 0.0) + now - t_phase[0]
        t_phase[0] = now
    if world == 1 and not args.force_comm:
        g = mesh.structured_box(n)
        plan = None
    else:
        g, plan = partition.slab_partition(n, n, n, world, rank)
    setup_breakdown = {"mesh_python": time.time() - t_setup}
    perm = None
    if args.ordering == "tile" and world == 1:
        perm = mesh.tile_ordering(n, n, n, 16, 16)
        g = mesh.permute_cells(g, perm)

    ctx = api.Context(local_rank)
    if args.nontemporal >= 0:
        ctx.set_option("nontemporal", args.nontemporal)
    for kv in args.opt:
        k_, v_ = kv.split("=")
        ctx.set_option(k_, int(v_))
    if transport == "rccl-plain":
        for k_ in RCCL_PLAIN_OPTIONS:
            ctx.set_option(k_, 0)
    connect(ctx)
    # RCCL's own account of the communicator, per rank (ncclCommCount / ncclCommUserRank / device / PCI bus id): a reader
    # of the line can check that RCCL saw N ranks on N distinct devices -- not what this script passed in
    rccl_view_all = None
    if world > 1 or args.force_comm:
        try:
            mine_ = dict(ctx.rccl_view(), rank=rank, local_rank=local_rank, transport=transport)
        except Exception as e:
            mine_ = {"rank": rank, "error": repr(e)[:300]}
        rccl_view_all = dist.all_gather_object(mine_)
    t_op = time.time()
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    setup_breakdown["operator_build"] = time.time() - t_op
    if plan is not None and plan.n_nbrs:
        mat.set_halo(plan.nbr_rank, plan.send_ptr, plan.send_idx, plan.recv_ptr)
    st = mat.stats()
    op = api.HipStencilOperator(mat, alpha=-1.0, beta=0.0)  # A = -L, SPD
    N = g.n_cells
    b = api.DeviceVector(ctx, N, g.n_halo)
    api.fill_with(b, 1.0)
    t_setup = time.time() - t_setup

    def run(iters: int, operator=None):
        x = api.DeviceVector(ctx, N, g.n_halo)
        s = api.CgSolver()
        s.num_iterations = iters
        s.absolute_error_tolerance = 0.0  # both tests disabled (Solver.hpp:136-139): exactly `iters` steps
        s.relative_error_tolerance = 0.0
        s.solve(x, b, operator if operator is not None else op)
        assert s.iteration == iters, (s.iteration, iters)
        return s, x

    def spmv_roofline(operator, stats, iters):
        """HIP-event pairs around every SpMV launch of one `iters`-iteration solve.  `achieved` / `frac` price the
        bytes the operator's record format really streams (records + x + y); for fp64 records those ARE SURVEY.md
        8d's algorithmic bytes.  The 8d figure over the same time is kept as `effective_vs_8d_GBs`."""
        ctx.set_option("profile_spmv", 1)
        run(iters, operator)
        samples = ctx.spmv_profile_samples()
        ctx.set_option("profile_spmv", 0)
        launches, total_ms, min_ms = int(samples.size), float(samples.sum()), float(samples.min())
        alg = 24 * N + 12 * stats["nnz_offdiag"]  # x + y + ext + (int32 col + f64 val) per entry
        fmt_bytes = stats["record_bytes"] + 16 * N  # the records this operator streams + x + y
        # One rank, tiled format-4 operator: from the second apply on, the SpMV kernel also ENDS the previous CG iteration
        # (x += alpha p, p' = r + beta p on the rows it loads: csrc/spmv.hip CgFuseArgs) -- it reads p, r, x and the
        # records and writes x, p', z: 48 B/row + records.  The solve's first apply is the plain kernel (and the
        # shortest launch of the set): the fused launches are priced on their own.
        fused = (world == 1 and not args.force_comm and stats.get("tiled_planes", 0) > 0 and launches == iters + 1 and
                 not any(kv.split("=")[0] == "cg_fuse" and int(kv.split("=")[1]) == 0 for kv in args.opt))
        if fused and launches > 1:
            rest = np.delete(samples, int(np.argmin(samples)))  # (the solve's first apply is the plain kernel)
            ms = (total_ms - min_ms) / (launches - 1)
            step_bytes = stats["record_bytes"] + 48 * N
            gbs = step_bytes / (ms * 1e-3) / 1e9
            return {"achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "bytes_per_launch": step_bytes, "fused_cg_step": True,
                    # SURVEY 8d's bytes of the reference statements this ONE launch covers: the apply (24 N + 12 nnz),
                    # x += alpha p (24 N), p = r + beta p (24 N), <p, z> (16 N)
                    "algorithmic_bytes_8d": alg, "bytes_8d_of_the_fused_statements": alg + 64 * N,
                    "effective_vs_8d_GBs": (alg + 64 * N) / (ms * 1e-3) / 1e9,
                    "frac_8d": (alg + 64 * N) / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "avg_launch_ms": ms, "median_launch_ms": float(np.median(rest)), "min_launch_ms": min_ms,
                    "frac_by_median": step_bytes / (float(np.median(rest)) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                    "launches_timed": launches}
        # one apply = one launch on a single GPU, an interior + a boundary launch on a partitioned mesh
        ms = total_ms / max(iters + 1, 1)
        gbs = fmt_bytes / (ms * 1e-3) / 1e9
        return {"achieved": gbs, "frac": gbs / HBM_PEAK_GBS, "bytes_per_launch": fmt_bytes, "fused_cg_step": False,
                "algorithmic_bytes_8d": alg, "effective_vs_8d_GBs": alg / (ms * 1e-3) / 1e9,
                "frac_8d": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "avg_launch_ms": ms, "median_launch_ms": float(np.median(samples)) * launches / max(iters + 1, 1),
                "min_launch_ms": min_ms, "launches_timed": launches}

    # Device spin-up (untimed, before the W warmup steps): the first process on an idle MI355X runs
    # ~20 % slow for its first several hundred milliseconds (clocks / memory power state); 10 ms of
    # warmup steps do not cover that.  Same work as the timed region, nothing is cached from it.
    # The loop count must be the same on every rank (each run() holds collectives), so the elapsed
    # time that decides it is the max over ranks.
    dist.barrier()  # every rank's operator is built: the first exchange's bounded waits start together
    t_spin = time.perf_counter()
    while args.spinup_seconds > 0 and dist.allreduce_max(time.perf_counter() - t_spin) < args.spinup_seconds:
        run(100)
        ctx.sync()
    if W > 0:
        run(W)
    ctx.sync()

    def timed_solve():
        """EXACTLY K steps between barrier + synchronize on both sides; max over ranks."""
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        s_, x_ = run(K)
        ctx.sync()
        torch.cuda.synchronize()
        local = time.perf_counter() - t0  # this rank's K steps are complete (every step holds two all-reduces)
        dist.barrier()
        return dist.allreduce_max(local), s_

    # The driver's `--steps 20` is 6 ms of work: one interval would decide the headline.  The K-step solve is
    # repeated until the timed repeats cover --min-seconds (same count on every rank: the elapsed times are
    # already max-reduced) and the median repeat is reported; `steps` stays K.
    IPC_KEYS = ("ipc_allreduce_wait_ticks", "ipc_allreduces", "ipc_ack_wait_ticks", "ipc_ack_waits",
                "ipc_halo_slow_poll_ticks", "ipc_halo_slow_polls")

    def ipc_counters():
        if world == 1 and not args.force_comm or transport != "ipc":
            return None
        try:
            return {k_: ctx.counter(k_) for k_ in IPC_KEYS}
        except Exception:
            return None

    phase("setup_spinup_warmup")
    repeats = []
    comm0 = ipc_counters()
    while True:
        e_, s = timed_solve()
        repeats.append(e_)
        if sum(repeats) >= args.min_seconds or len(repeats) >= 2000:
            break
    comm1 = ipc_counters()
    elapsed = float(np.median(repeats))
    final_residual = s.absolute_error
    # N > 1: where the communication time of an iteration went, from the device's own clock (peer-window transport: the
    # kernels time their waits, csrc/ipc_device.hpp IpcDev::stat); the worst rank's numbers are reported
    comm_breakdown = None
    if world > 1 or args.force_comm:
        if comm0 is not None and comm1 is not None:
            d_ = {k_: comm1[k_] - comm0[k_] for k_ in IPC_KEYS}
            its_ = max(len(repeats) * K, 1)
            mine = [d_["ipc_allreduce_wait_ticks"] * 0.01 / its_, d_["ipc_ack_wait_ticks"] * 0.01 / its_,
                    d_["ipc_halo_slow_poll_ticks"] * 0.01 / max(d_["ipc_halo_slow_polls"], 1), d_["ipc_halo_slow_polls"] / its_]
            worst = [dist.allreduce_max(v_) for v_ in mine]
            comm_breakdown = {
                "transport": "ipc", "iterations_covered": its_,
                "allreduce_wait_us_per_iteration_worst_rank": worst[0],
                "allreduces_per_iteration": d_["ipc_allreduces"] / its_,
                "allreduce_wait_us_each_this_rank": d_["ipc_allreduce_wait_ticks"] * 0.01 / max(d_["ipc_allreduces"], 1),
                "send_ack_wait_us_per_iteration_worst_rank": worst[1],
                "halo_values_not_there_at_first_look_per_iteration_worst_rank": worst[3],
                "halo_late_value_mean_wait_us_worst_rank": worst[2],
                "note": "all-reduce wait = from a rank's own contribution being stored into every window until every rank's "
                        "has been read from its own (link latency + the slowest rank's lead); send wait = a sending block "
                        "waiting for the receivers' acknowledgement of the plane two exchanges back; late halo values = "
                        "rows of the boundary launch that had to poll (thread-time each); everything else of an iteration "
                        "is the single-GPU path's kernels (compare ms_per_step with the 1-GPU line)"}
        elif transport in ("rccl", "rccl-plain") and not args.shared_device:
            # RCCL: an instrumented solve beside the timed ones (stamp kernels around every step of the exchange and the
            # all-reduces, csrc/comm.hip); the worst rank's figures
            try:
                ctx.set_option("profile_comm", 1)
                run(K)
                ctx.sync()
                comm_breakdown = ctx.rccl_profile(K)
                ctx.set_option("profile_comm", 0)
                for k_ in list(comm_breakdown):
                    if k_.endswith(("_each", "_per_iteration")) and k_ not in ("halo_exchanges_per_iteration", "allreduces_per_iteration"):
                        comm_breakdown[k_ + "_worst_rank"] = dist.allreduce_max(comm_breakdown.pop(k_))
            except Exception as e:
                comm_breakdown = {"transport": transport, "error": repr(e)}
        else:
            comm_breakdown = {"transport": transport,
                              "note": "host-staged transport: synchronous, nothing to overlap; compare ms_per_step with the 1-GPU line"}

    phase("timed_region_and_breakdown")
    # ---- --contact: BASELINE configs[2]'s solver once over the same transport, with its breakdown ----
    contact_bicgstab = None
    if args.contact and (world > 1 or args.force_comm):
        try:
            def run_bicg(iters):
                x_ = api.DeviceVector(ctx, N, g.n_halo)
                s_ = api.BiCgStabSolver()
                s_.num_iterations, s_.absolute_error_tolerance, s_.relative_error_tolerance = iters, 0.0, 0.0
                s_.solve(x_, b, op)
                return s_

            run_bicg(max(W, 3))
            ctx.sync()
            dist.barrier()
            t0 = time.perf_counter()
            sb_ = run_bicg(K)
            ctx.sync()
            tb_ = dist.allreduce_max(time.perf_counter() - t0)
            contact_bicgstab = {"iterations": sb_.iteration, "us_per_iteration": tb_ / K * 1e6, "final_residual": sb_.absolute_error}
            if transport in ("rccl", "rccl-plain") and not args.shared_device:
                ctx.set_option("profile_comm", 1)
                run_bicg(K)
                ctx.sync()
                br_ = ctx.rccl_profile(K)
                ctx.set_option("profile_comm", 0)
                contact_bicgstab["comm_breakdown"] = {k_: (dist.allreduce_max(v_) if isinstance(v_, float) else v_)
                                                      for k_, v_ in br_.items() if k_ != "note"}
        except Exception as e:
            contact_bicgstab = {"error": repr(e)[:300]}

    # ---- N > 1: post-flight.  The timed region ran the production kernels on the production transport; before its
    # number is reported, (1) the fused CG step must agree with the kernel-per-statement loop on the same transport
    # (same K iterations, residual to 1e-9) and (2) ONE apply of the full-size operator to x = global plane index must
    # vanish on every row away from the walls -- which fails iff a halo plane is stale or misplaced.  Any failure ends
    # this process with an error: the supervisors then start fresh ranks on the next transport.
    postflight = None
    if world > 1:
        ctx.set_option("cg_fuse", 0)
        s_ref, _ = run(K)
        ctx.set_option("cg_fuse", 1)
        res_diff = abs(s_ref.absolute_error - final_residual) / final_residual
        kz = (np.asarray(g.global_id[:N], dtype=np.int64) // (n * n)).astype(np.float64)
        xz = api.DeviceVector(ctx, N, g.n_halo)
        xz.upload(kz)
        yz = api.DeviceVector(ctx, N, g.n_halo)
        op.mul(yz, xz)
        gid = np.asarray(g.global_id[:N], dtype=np.int64)
        ii, jj, kk = gid % n, (gid // n) % n, gid // (n * n)
        away = (ii > 0) & (ii < n - 1) & (jj > 0) & (jj < n - 1) & (kk > 0) & (kk < n * world - 1)
        # (a wrong plane shows as +-1 times a face weight ~ n^2; spacings that are not exact in binary leave ~1e-16 n^3)
        bad_rows = int(np.count_nonzero(np.abs(yz.to_numpy()[away]) > 1e-6 * n * n))
        ok_all = dist.allreduce_max(0.0 if (bad_rows == 0 and res_diff <= 1e-9 and np.isfinite(final_residual) and
                                            not inject_post) else 1.0) == 0.0
        postflight = {"ok": ok_all, "halo_rows_wrong_on_this_rank": bad_rows,
                      "fused_vs_unfused_residual_rel_diff": res_diff, "steps_compared": K}
        del xz, yz
        if not ok_all:
            print(f"bench.py: post-flight FAILED on rank {rank}: {postflight}", file=sys.stderr, flush=True)
            dist.barrier()
            os._exit(43)

    # ---- roofline of the SpMV: HIP-event pairs around every launch --------------------------------
    march_planes = 8  # the library's default for option cg_march (csrc/common.hpp)
    for kv in args.opt:
        if kv.split("=")[0] == "cg_march":
            march_planes = int(kv.split("=")[1])
    prof_iters = max(K, args.roofline_launches)  # (a sample of 21 launches moved the fraction by 0.07 between runs)
    roof = spmv_roofline(op, st, prof_iters)
    phase("postflight_and_roofline_launches")
    # The CPU baseline's samples are pure host work in ONE thread (ctypes calls: the interpreter lock is released): they run
    # beside the GPU-side blocks that follow (child processes, stress variants, BASELINE configs) instead of after them --
    # 15 s of the run's wall time.  None of those blocks is the headline, which is complete at this point.
    cpu_thread, cpu_box = None, {}
    if rank == 0 and world == 1 and args.cpu_iters > 0:
        import threading

        def _cpu_work():
            try:
                cpu_box["samples"] = cpu_samples(args, n, g, perm)
            except Exception as e:  # the baseline must never cost the headline line
                cpu_box["error"] = repr(e)

        cpu_thread = threading.Thread(target=_cpu_work, name="cpu_baseline", daemon=True)
        cpu_thread.start()
    fmt_name = record_format_name(st)
    # HBM bytes per launch by PMC: measured by a child of THIS run (two rocprofv3 --pmc passes over a short solve of
    # the same operator), else quoted from the committed profile -- under its own name, with the file's hash
    tfile = os.path.join(ROOT, "profiles", "spmv_hbm_traffic.json")
    traffic, traffic_general, traffic_note, traffic_plain = None, None, None, {}
    tet_dir, tet_prefix, tet_file_seconds, tet_file_bytes = None, None, None, None
    if world == 1 and not args.force_comm and not args.skip_unstructured3d:
        try:
            import tempfile

            tet_dir = tempfile.mkdtemp(prefix="storm_tet_", dir=os.environ.get("TMPDIR", "/tmp"))
            tet_prefix, tet_file_seconds, tet_file_bytes = tet_files(args.tet_edge, tet_dir)
        except Exception as e:
            tet_prefix, tet_file_seconds = None, {"error": repr(e)}
    if args.traffic == "measure" and n == 256 and world == 1 and rank == 0:
        ctx.sync()
        phase("tet_files")
        traffic, traffic_general, traffic_note, traffic_plain = measure_traffic(args, tet_prefix)
        phase("pmc_traffic_children")
    # ---- the SpMV alone: >= 50 stand-alone applies on x_i = sin(0.37 i), median, three byte counts ----
    spmv_block = None
    if world == 1 and not args.force_comm and not args.skip_spmv:
        try:
            cell_ids = np.arange(N) if perm is None else perm
            spmv_block = {"lattice" if st["value_dictionary_size"] else "general":
                          spmv_standalone(api, ctx, mat, st, cell_ids, args.spmv_launches, traffic_plain.get("lattice" if st["value_dictionary_size"] else "general"))}
        except Exception as e:
            spmv_block = {"error": repr(e)}
    traffic_from_profile = None
    if args.traffic != "off" and os.path.exists(tfile) and n == 256 and world == 1:
        try:
            import hashlib

            raw = open(tfile, "rb").read()
            tj = json.loads(raw)
            traffic_from_profile = {"file": "profiles/spmv_hbm_traffic.json", "sha256": hashlib.sha256(raw).hexdigest(),
                                    "bytes_per_launch": tj.get("traffic_bytes_per_launch") if tj.get("record_format") == fmt_name else None,
                                    "general_bytes_per_launch": tj.get("general", {}).get("traffic_bytes_per_launch"),
                                    "note": "from a committed rocprofv3 --pmc profile of an earlier run, NOT measured in this run"}
        except Exception:
            traffic_from_profile = None

    phase("spmv_standalone_lattice")
    # ---- the same problem through the fp64 records (what a mesh with all-distinct weights gets) ----
    general = None
    general_roof = None
    if world == 1 and not args.skip_general:
        try:
            if st["value_dictionary_size"]:
                ctx.set_option("spmv_dict", 0)
                mat0 = api.StencilMatrix.from_face_graph(ctx, g)
                ctx.set_option("spmv_dict", 4)
            else:
                mat0 = mat
            op0 = api.HipStencilOperator(mat0, alpha=-1.0, beta=0.0)
            run(max(W, 20), op0)
            ctx.sync()
            reps0 = []
            while sum(reps0) < args.min_seconds and len(reps0) < 2000:
                t1 = time.perf_counter()
                run(K, op0)
                ctx.sync()
                reps0.append(time.perf_counter() - t1)
            t1 = float(np.median(reps0))
            r0 = spmv_roofline(op0, mat0.stats(), prof_iters)
            general = {"record_format": "fp64 weights + int32 columns", "cg_iter_per_s": K / t1,
                       "ms_per_step": t1 / K * 1e3, "repeats": len(reps0)}
            gt = traffic_general
            general_roof = {"kernel": "spmv_sell_kernel (fp64 records: the format any mesh gets) + fused <p,Ap> partials",
                            "bound": "hbm", "achieved": r0["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": r0["frac"], "frac_8d": r0.get("frac_8d"), "traffic": gt, "bytes_per_launch": r0["bytes_per_launch"],
                            "algorithmic_bytes_8d": r0["algorithmic_bytes_8d"], "avg_launch_ms": r0["avg_launch_ms"],
                            "min_launch_ms": r0["min_launch_ms"], "launches_timed": r0["launches_timed"]}
            if mat0 is not mat and isinstance(spmv_block, dict) and not args.skip_spmv:
                try:
                    spmv_block["general"] = spmv_standalone(api, ctx, mat0, mat0.stats(), np.arange(N) if perm is None else perm,
                                                            args.spmv_launches, traffic_plain.get("general"))
                except Exception as e:
                    spmv_block["general"] = {"error": repr(e)}
            if mat0 is not mat:
                mat0.close()
        except Exception as e:
            general = {"error": repr(e)}

    # ---- SURVEY.md 8d's unstructured stress variant: the cells renumbered by the seeded permutation, then the library's
    # ordering -- all on the library's host meshes (storm_hip_mesh_*: threaded; round 4 did this in numpy: 7.4 s) ----
    phase("general_records")
    permuted, unstructured = None, None
    stress_ready = world == 1 and not args.skip_general and not (args.skip_permuted and args.skip_unstructured)
    if stress_ready:
        from stormruler_amd import host_mesh

        g0 = g if perm is None else mesh.structured_box(n)
        perm0 = mesh.random_permutation(N)

    def stress_variant(graph, mode, reference_residual):
        """Scramble, order (`mode`), build, time and profile one variant of the problem; the residual after K iterations must
        be the natural order's (the same operator conjugated by a permutation)."""
        sec = {}
        t0 = time.time()
        hm = host_mesh.HostMesh.from_face_graph(graph)
        hm.permute_cells(perm0)
        sec["copy_and_scramble"] = time.time() - t0
        t0 = time.time()
        kind = hm.order_cells(mode)
        sec["ordering"] = time.time() - t0
        t0 = time.time()
        matp = hm.create_operator(ctx)
        sec["operator_build"] = time.time() - t0
        stp = matp.stats()
        opp = api.HipStencilOperator(matp, alpha=-1.0, beta=0.0)
        run(max(W, 20), opp)
        ctx.sync()
        repsp = []
        while sum(repsp) < args.min_seconds and len(repsp) < 2000:
            t1 = time.perf_counter()
            sp_, _ = run(K, opp)
            ctx.sync()
            repsp.append(time.perf_counter() - t1)
        t1 = float(np.median(repsp))
        rp = spmv_roofline(opp, stp, prof_iters)
        v_ = hm.view()
        band = np.abs(np.ctypeslib.as_array(v_.inner, shape=(v_.n_faces,)) - np.ctypeslib.as_array(v_.outer, shape=(v_.n_faces,)))
        out_ = {"kernel": kernel_name(stp), "record_format": record_format_name(stp), "bound": "hbm",
                "achieved": rp["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": rp["frac"], "frac_8d": rp.get("frac_8d"), "traffic": None,
                "bytes_per_launch": rp["bytes_per_launch"], "algorithmic_bytes_8d": rp["algorithmic_bytes_8d"],
                "effective_vs_8d_GBs": rp["effective_vs_8d_GBs"], "avg_launch_ms": rp["avg_launch_ms"],
                "median_launch_ms": rp.get("median_launch_ms"), "min_launch_ms": rp["min_launch_ms"],
                "launches_timed": rp["launches_timed"], "cg_iter_per_s": K / t1, "ms_per_step": t1 / K * 1e3,
                "final_residual_rel_diff_vs_natural_order": abs(sp_.absolute_error - reference_residual) / reference_residual,
                "max_column_distance": int(band.max()), "ordering_kind": kind, "host_seconds": sec,
                "host_seconds_permute_order_build": sum(sec.values())}
        matp.close()
        hm.close()
        return out_

    if stress_ready and not args.skip_permuted:
        try:
            # the library's ordering (storm_hip_order_cells): the lexicographic order of a lattice where the cell centres form
            # one -- the scrambled box gets its natural order, and the lattice records, back
            permuted = stress_variant(g0, "auto", final_residual)
            permuted["ordering"] = ("numpy.random.default_rng(12345).permutation(N) (storm_hip_mesh_permute_cells), then the "
                                    f"library's ordering from the cell centres (storm_hip_order_cells, mode auto): {permuted['ordering_kind']}")
        except Exception as e:
            permuted = {"error": repr(e)}
    # ---- ... and the same with a jittered geometry: no two weights equal, so the records are fp64 weights + int32
    # columns (SURVEY.md 8d's bytes) AND the ordering is FORCED onto the Z-order curve (a Triangle / TetGen mesh has no
    # lattice to find)
    if stress_ready and not args.skip_unstructured:
        try:
            tj = time.time()
            gj = mesh.jitter_geometry(g0, 1.0 / n)
            tj = time.time() - tj
            matj = api.StencilMatrix.from_face_graph(ctx, gj)
            sj, _ = run(K, api.HipStencilOperator(matj, alpha=-1.0, beta=0.0))  # the natural order's residual
            matj.close()
            unstructured = stress_variant(gj, "morton", sj.absolute_error)
            del gj
            unstructured["host_seconds"]["jitter_geometry_numpy"] = tj
            unstructured["geometry"] = ("cell centres displaced by <= 0.2 h per coordinate, face areas scaled by 1 +- 0.1 "
                                        "(numpy.random.default_rng(2024)): all weights distinct")
            unstructured["ordering"] = ("the same seeded permutation, then the Z-order (Morton) curve of the cell centres "
                                        "(storm_hip_order_cells, mode morton)")
        except Exception as e:
            unstructured = {"error": repr(e)}

    phase("stress_variants")
    # ---- a genuinely unstructured 3-D mesh: tetrahedra from TetGen files (variable row degree, fp64 records) ----
    unstructured3d = None
    if tet_prefix is not None:
        try:
            unstructured3d = tet_variant(api, ctx, args, tet_prefix, tet_file_seconds, tet_file_bytes,
                                         traffic_plain.get("tet"), traffic_plain.get("tet_dot"))
        except Exception as e:
            unstructured3d = {"error": repr(e)}
    elif isinstance(tet_file_seconds, dict) and "error" in tet_file_seconds:
        unstructured3d = tet_file_seconds
    if tet_dir is not None:
        import shutil

        shutil.rmtree(tet_dir, ignore_errors=True)

    phase("tetrahedra")
    # ---- BASELINE configs 3, 4, 5 on this GPU (bounded: a few hundred milliseconds of device time each) ----
    configs = None
    if world == 1 and not args.skip_configs and not args.force_comm:
        try:
            configs = baseline_configs(api, mesh, ctx, op, b, N, n, st, args.min_seconds,
                                       tuple(v for v in args.self_exchange.split(",") if v))
        except Exception as e:
            configs = {"error": repr(e)}

    phase("baseline_configs_and_children")
    # a measured device-copy ceiling in the same run (achievable HBM rate, for context)
    # (two 1 GiB buffers: far beyond the 256 MiB Infinity Cache, so this is an HBM number)
    copy_gbs = None
    try:
        nc = 1 << 27
        src = api.DeviceVector(ctx, nc)
        dst = api.DeviceVector(ctx, nc)
        for _ in range(2):
            dst <<= src
        ctx.timer_start()
        reps = 10
        for _ in range(reps):
            dst <<= src
        copy_gbs = 16.0 * nc * reps / (ctx.timer_stop() * 1e-3) / 1e9
        del src, dst
    except Exception:
        pass

    # ---- BLAS-1 rates at this size ("plus BLAS-1 GB/s per kernel") ------------------
    blas1 = None
    if world == 1 and not args.skip_blas1:  # (dots on a context with a communicator are collective calls)
        try:
            blas1 = blas1_rates(api, ctx, N)
        except Exception as e:
            blas1 = {"error": repr(e)}

    phase("copy_and_blas1")
    # ---- CPU baseline: the oracle (port of the reference path), 1 thread, bounded sample -------
    cpu = None
    if cpu_thread is not None:
        cpu_thread.join()
        try:
            if "error" in cpu_box:
                cpu = {"error": cpu_box["error"]}
            else:
                cpu = cpu_baseline(args, n, g, perm, run, ctx, cpu_box["samples"])
                cpu["timed_beside"] = ("the GPU-side blocks of this run (PMC child processes, stress variants, BASELINE configs): one "
                                       "host thread of this process, the GPU work of those blocks on other cores")
        except Exception as e:  # the baseline must never cost the headline line
            cpu = {"error": repr(e)}

    phase("cpu_baseline")
    # what a pure streaming kernel with the SpMV's read:write mix (two 8-byte reads, one 8-byte write per row) reaches on
    # this chip in the same run: `a <<= b - c` over three distinct 134 MB vectors
    mix_ceiling = None
    if isinstance(blas1, dict) and isinstance(blas1.get("sub (a <<= b - c)"), dict):
        mix_ceiling = blas1["sub (a <<= b - c)"]["GBs"]

    if rank == 0:
        value = world * K / elapsed
        out = {
            "metric": "CG iterations/sec, 256^3 Poisson per GPU (+ SpMV achieved HBM GB/s in `roofline`)",
            "value": value,
            "unit": "iter/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"3D 7-point Poisson {n}^3 per GPU (structured-as-unstructured FVM face graph), fp64 CG, "
                            f"no preconditioner, b=1, x0=0 [BASELINE.json configs[1]]",
                "cells_per_gpu": N, "interior_faces_per_gpu": g.n_faces, "ordering": args.ordering,
                "partition": "single GPU" if world == 1 else
                             (("DEBUG, ranks share ONE device: " if args.shared_device else "") + f"z-slabs, {world} ranks, " +
                              {"ipc": "peer-window halo + all-reduce (hipIpc-mapped windows, direct stores over xGMI)",
                               "rccl": "RCCL halo send/recv + all-reduce over xGMI",
                               "rccl-plain": "RCCL halo send/recv + all-reduce over xGMI, the plain form (cross-stream events, "
                                             "two-launch reductions, no early halo, no fused step)",
                               "host": "halo planes and scalars staged through host memory (gloo)"}[transport]),
                "value_definition": "n_gpus x K / max-over-ranks wall time of a K-iteration solve (init residual included; "
                                    "barrier + synchronize, clock, K steps, synchronize, clock, barrier); median over the "
                                    "repeats listed in `timing`",
            },
            "roofline": {
                "kernel": (("cg_step_march_kernel (one CG step per launch: x += alpha p, p' = r + beta p, z = A p', <p',z> partials; "
                            "blocks of 1024 rows marching through the planes)" if march_planes > 0 else
                            "spmv_canon_tile_kernel<FUSE> (one CG step per launch: x += alpha p, p' = r + beta p, z = A p', "
                            "<p',z> partials; tiles of 1024 rows x %d planes)" % st["tiled_planes"]) if roof.get("fused_cg_step") else
                           ("spmv_canon_tile_kernel (tiles of 1024 rows x %d planes) + fused <p,Ap> partials" % st["tiled_planes"])
                           if st.get("tiled_planes") else kernel_name(st) + " (sliced-ELL gather SpMV + fused <p,Ap> partials)"),
                "fused_cg_step": roof.get("fused_cg_step"),
                "bound": "hbm", "achieved": roof["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": roof["frac"], "frac_8d": roof.get("frac_8d"),
                "bytes_8d_of_the_fused_statements": roof.get("bytes_8d_of_the_fused_statements"),
                "traffic": traffic, "traffic_method": traffic_note,
                "traffic_from_profile": traffic_from_profile,
                "bytes_per_launch": roof["bytes_per_launch"], "record_format": fmt_name,
                "avg_launch_ms": roof["avg_launch_ms"], "median_launch_ms": roof.get("median_launch_ms"),
                "frac_by_median": roof.get("frac_by_median"), "min_launch_ms": roof["min_launch_ms"],
                "launches_timed": roof["launches_timed"], "measured_copy_GBs": copy_gbs,
                "measured_2read_1write_stream_GBs": mix_ceiling,
                "frac_of_measured_stream": (roof["achieved"] / mix_ceiling) if mix_ceiling else None,
                "algorithmic_bytes_8d": roof["algorithmic_bytes_8d"],
                "effective_vs_8d_GBs": roof["effective_vs_8d_GBs"],
                "note": "achieved/frac = bytes the dominant kernel streams per launch (plain SpMV: records + x + y; fused CG "
                        "step: records + p, r, x read + x, p', z written; `traffic` = the same by PMC) / launch time: a "
                        "physical HBM fraction.  frac_8d / effective_vs_8d_GBs divide SURVEY 8d's bytes of the reference "
                        "statements the launch covers (fp64 weights + int32 columns; the fused step: + 64 B/row of vector "
                        "statements) by the same time: > 1 for the lossless byte-indexed lattice format, which does not move "
                        "those bytes -- NOT a bandwidth, and it says nothing about a mesh with distinct weights.  The SpMV "
                        "alone is the `spmv` block (stand-alone applies, median of >= 50); roofline_general / "
                        "roofline_unstructured3d are the fp64-record kernel every mesh can use, where frac == frac_8d",
            },
            "spmv": spmv_block,
            "roofline_general": general_roof,
            "roofline_permuted_rcm": permuted,
            "roofline_unstructured": unstructured,
            "roofline_unstructured3d": unstructured3d,
            "config1_cg64": (configs or {}).get("config1_cg64") if isinstance(configs, dict) else None,
            "config3_bicgstab256": (configs or {}).get("config3_bicgstab256") if isinstance(configs, dict) else None,
            "config4_gmres30_convdiff128": (configs or {}).get("config4_gmres30_convdiff128") if isinstance(configs, dict) else None,
            "config5_cavity128": (configs or {}).get("config5_cavity128") if isinstance(configs, dict) else None,
            "extra_gmres30_poisson256": (configs or {}).get("extra_gmres30_poisson256") if isinstance(configs, dict) else None,
            "host_loop_cg256": (configs or {}).get("host_loop_cg256") if isinstance(configs, dict) else None,
            "multi_rank_path_at_one_rank": (configs or {}).get("multi_rank_path_at_one_rank") if isinstance(configs, dict) else None,
            "configs_error": configs.get("error") if isinstance(configs, dict) else None,
            "value_general": general.get("cg_iter_per_s") if isinstance(general, dict) else None,
            "general_mesh_path": general,
            "blas1": blas1,
            "cpu_baseline": cpu,
            "timing": {"repeats": len(repeats), "timed_seconds_total": float(sum(repeats)),
                       "ms_per_step_min": min(repeats) / K * 1e3, "ms_per_step_max": max(repeats) / K * 1e3,
                       "ms_per_step_median": elapsed / K * 1e3},
            "cg": {"iterations_per_sec_global": K / elapsed,
                   "algorithmic_bytes_per_iteration": roof["algorithmic_bytes_8d"] + 96 * N,
                   "reference_op_list_bytes_over_time_GBs_NOT_A_BANDWIDTH": (roof["algorithmic_bytes_8d"] + 96 * N) * K / elapsed / 1e9,
                   "bytes_really_moved_per_iteration": roof["bytes_per_launch"] + (24 if roof.get("fused_cg_step") else 64) * N,
                   "achieved_GBs_bytes_really_moved": (roof["bytes_per_launch"] + (24 if roof.get("fused_cg_step") else 64) * N) * K / elapsed / 1e9,
                   "final_residual": final_residual},
            "op_stats": st,
            "device": ctx.info()["name"],
            "setup_seconds": t_setup,
            "setup_breakdown_seconds": setup_breakdown,
            "phase_seconds": phases,
        }
        if world > 1 or args.force_comm:
            out["transport"] = transport
        if world > 1:
            out["preflight"] = preflight
            out["postflight"] = postflight
        if comm_breakdown is not None:
            out["comm_breakdown"] = comm_breakdown
        mr = out.get("multi_rank_path_at_one_rank")
        if isinstance(mr, dict) and isinstance(mr.get("cg"), dict):  # against this run's own one-rank step
            mr["cg"]["us_per_iteration_plain"] = out["ms_per_step"] * 1e3
            mr["cg"]["overhead_us_per_iteration"] = mr["cg"]["us_per_iteration_over_rccl"] - out["ms_per_step"] * 1e3
        if world > 1 or args.force_comm:
            out["rccl_view"] = rccl_view_all
        if contact_bicgstab is not None:
            out["contact_bicgstab"] = contact_bicgstab
        if os.environ.get("STORM_BENCH_WORKER") == "1":
            # one rank of an N > 1 run: the whole record goes to the supervisor (a file of its own, not the driver's
            # stdout), which merges the transports and prints the compact line
            print(json.dumps(out), flush=True)
        else:
            emit(out, args.detail_path)
    dist.barrier()
    try:  # leave no dangling process group / communicator behind
        mat.close()
        ctx.sync()
        dist.barrier()  # nobody frees a peer window another rank's kernels may still write to
        ctx.close()
        import torch.distributed as td

        if td.is_available() and td.is_initialized():
            td.destroy_process_group()
    except Exception:
        pass
    return 0


LINE_CAP = 8000  # the driver keeps ~8 KB of stdout tail: a longer line is not parsed (BENCH_r05.json: parsed = null)


def _num(v, digits=6):
    """Floats to `digits` significant digits (the line is a summary; bench_detail.json keeps every digit)."""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float(f"{v:.{digits}g}")
    if isinstance(v, dict):
        return {k: _num(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_num(x, digits) for x in v]
    return str(v)


def _get(d, *path):
    for k in path:
        if not isinstance(d, dict) or k not in d:
            return None
        d = d[k]
    return d


def _ratio(a, b):
    try:
        return float(a) / float(b) if a is not None and b else None
    except (TypeError, ValueError, ZeroDivisionError):
        return None


def _short(text, n):
    text = "" if text is None else str(text)
    return text if len(text) <= n else text[: n - 3] + "..."


def compact_record(full: dict) -> dict:
    """The ONE line the driver parses: the contract's fields, `roofline` (dominant kernel + the SURVEY-8d-comparable
    fp64-record SpMV figures inside it), `cpu_baseline`, `value_general`, one or two numbers per BASELINE config, and for
    N > 1 the transports, RCCL's own view of the ranks and the breakdown's numbers.  Everything else -- every block of the
    full record, with its notes -- is in bench_detail.json (`detail`)."""
    r = full.get("roofline") or {}
    sg = _get(full, "spmv", "general") or {}
    sl = _get(full, "spmv", "lattice") or {}
    u3 = full.get("roofline_unstructured3d") or {}
    cfg = full.get("config") or {}
    roof = {
        "kernel": _short(r.get("kernel"), 60).split(" (")[0], "bound": r.get("bound"), "achieved": r.get("achieved"), "peak": r.get("peak"),
        "unit": r.get("unit"), "frac": r.get("frac"), "traffic": r.get("traffic"),
        "bytes_per_launch": r.get("bytes_per_launch"), "avg_launch_ms": r.get("avg_launch_ms"),
        "launches_timed": r.get("launches_timed"), "frac_8d": r.get("frac_8d"),
        "record_format": _short(r.get("record_format"), 80),
        # SURVEY 8d's own figure: the SpMV alone on fp64 records (streamed bytes == 8d's algorithmic bytes)
        "spmv_general_kernel": sg.get("kernel"),
        "spmv_general_bytes_8d": sg.get("algorithmic_bytes_8d"),
        "spmv_general_ms": _get(sg, "rotating_3_pairs", "median_ms"),
        "spmv_general_frac_8d_rotating": _get(sg, "rotating_3_pairs", "frac_8d"),
        "spmv_general_frac_8d_back_to_back": _get(sg, "back_to_back", "frac_8d"),
        "spmv_general_traffic": sg.get("pmc_traffic_bytes"),
        "spmv_lattice_ms": _get(sl, "rotating_3_pairs", "median_ms"),
        "spmv_lattice_frac_streamed_rotating": _get(sl, "rotating_3_pairs", "frac_streamed"),
        "spmv_lattice_traffic": sl.get("pmc_traffic_bytes"),
        "cg_general_frac_8d": _get(full, "roofline_general", "frac_8d"),
        "tets_frac_8d": u3.get("frac_8d"), "tets_spmv_frac_8d_rotating": _get(u3, "spmv", "rotating_3_pairs", "frac_8d"),
        "tets_traffic_over_8d": u3.get("traffic_over_8d_bytes"),
        "measured_copy_GBs": r.get("measured_copy_GBs"),
        # "report both fractions" -- of the 8 TB/s peak (frac, spmv_general_frac_8d_rotating) and of what a
        # device copy reaches in this very run
        "frac_of_measured_copy": _ratio(r.get("achieved"), r.get("measured_copy_GBs")),
        "spmv_general_frac_of_measured_copy": _ratio(_get(sg, "rotating_3_pairs", "GBs_8d_bytes"), r.get("measured_copy_GBs")),
        "note": "frac: streamed bytes of the dominant kernel / time / peak; frac_8d > 1 on the lattice records is not a "
                "bandwidth; spmv_general_*: stand-alone fp64-record SpMV, SURVEY 8d bytes, median of >= 50",
    }
    cpu_full = full.get("cpu_baseline")
    cpu = None
    if isinstance(cpu_full, dict):
        cpu = {k: cpu_full.get(k) for k in ("value", "unit", "cores", "kind") if k in cpu_full}
        cpu["sample"] = _short(cpu_full.get("sample"), 200)
        for k in ("value_fma_build", "value_native_O3", "value_native_O3_fast_math", "native_flags", "gpu_vs_cpu_residual_rel_diff", "error"):
            if cpu_full.get(k) is not None:
                cpu[k] = _short(cpu_full[k], 120) if isinstance(cpu_full[k], str) else cpu_full[k]
        par = cpu_full.get("parallel")
        if isinstance(par, dict) and "value" in par:
            cpu["openmp_value"], cpu["openmp_cores"] = par.get("value"), par.get("cores")
            if par.get("value_min") is not None:
                cpu["openmp_value_min"] = par.get("value_min")
        c1 = cpu_full.get("config1_64cubed")
        if isinstance(c1, dict):
            cpu["config1_64cubed"] = {k: c1.get(k) for k in ("cpu_iterations", "gpu_iterations", "cpu_seconds", "gpu_seconds", "solution_rel_diff")}
    line = {
        "metric": full.get("metric"), "value": full.get("value"), "unit": full.get("unit"), "n_gpus": full.get("n_gpus"),
        "steps": full.get("steps"), "warmup": full.get("warmup"), "ms_per_step": full.get("ms_per_step"),
        "higher_is_better": full.get("higher_is_better"), "scaling": full.get("scaling"), "vs_baseline": full.get("vs_baseline"),
        "dtype": full.get("dtype"), "data": full.get("data"),
        "config": {"workload": _short(cfg.get("workload"), 220), "cells_per_gpu": cfg.get("cells_per_gpu"),
                   "partition": _short(cfg.get("partition"), 160), "ordering": cfg.get("ordering")},
        "roofline": roof if full.get("roofline") is not None else None, "cpu_baseline": cpu, "value_general": full.get("value_general"),
    }
    if full.get("error"):  # (a run that measured nothing says why: failure_record)
        line["error"] = _short(full["error"], 600)
    configs = {}
    for key, fields in (("config1_cg64", ("us_per_iteration",)), ("config3_bicgstab256", ("us_per_iteration", "frac")),
                        ("config4_gmres30_convdiff128", ("us_per_inner_iteration", "frac")),
                        ("config5_cavity128", ("s_per_step",)), ("extra_gmres30_poisson256", ("us_per_inner_iteration", "frac")),
                        ("host_loop_cg256", ("host_loop_lazy_over_device_loop",))):
        blk = full.get(key)
        if isinstance(blk, dict):
            configs[key] = {"error": _short(blk["error"], 100)} if "error" in blk else {f: blk.get(f) for f in fields}
    mr = full.get("multi_rank_path_at_one_rank")
    if isinstance(mr, dict):
        configs["multi_rank_path_at_one_rank"] = ({"error": _short(mr["error"], 100)} if "error" in mr else
                                                  {s_: _get(mr, s_, "overhead_us_per_iteration") for s_ in ("cg", "bicgstab") if s_ in mr})
    if configs:
        line["configs"] = configs
    b1 = full.get("blas1")
    if isinstance(b1, dict) and "error" not in b1:
        line["blas1_frac"] = {k.split(" (")[0].replace(",", "").replace(" ", "_"): v.get("frac_of_peak") for k, v in b1.items() if isinstance(v, dict)}
    if isinstance(full.get("timing"), dict):
        line["timing"] = {k: full["timing"].get(k) for k in ("repeats", "ms_per_step_min", "ms_per_step_median", "ms_per_step_max")}
    line["final_residual"] = _get(full, "cg", "final_residual")
    line["device"] = full.get("device")
    if (full.get("n_gpus") or 1) > 1 or full.get("rccl_view") is not None:
        line["transport"] = full.get("transport")
        rv = full.get("rccl_view")
        if isinstance(rv, list):
            counts = sorted({v.get("nccl_comm_count") for v in rv if isinstance(v, dict) and "nccl_comm_count" in v})
            line["rccl"] = {"nccl_comm_count": counts[0] if len(counts) == 1 else counts,
                            "distinct_devices": len({v.get("pci_bus_id") for v in rv if isinstance(v, dict) and v.get("pci_bus_id")}),
                            "ranks": [[v.get("rank"), v.get("nccl_user_rank"), v.get("hip_device"), v.get("pci_bus_id")]
                                      if isinstance(v, dict) and "error" not in v else [_get(v, "rank"), _short(_get(v, "error"), 60)]
                                      for v in rv][:16],
                            "ranks_columns": "rank, ncclCommUserRank, hip device, pci bus id"}
        tm = full.get("transports_measured")
        if isinstance(tm, dict):
            line["transports_measured"] = {
                t: {"value": m.get("value"), "ms_per_step": m.get("ms_per_step"), "postflight_ok": _get(m, "postflight", "ok"),
                    "comm_breakdown": _numbers_only(m.get("comm_breakdown"))} for t, m in tm.items()}
        elif full.get("comm_breakdown") is not None:
            line["comm_breakdown"] = _numbers_only(full.get("comm_breakdown"))
        fb = full.get("transport_fallback")
        if fb:
            line["transport_fallback"] = [{"transport": f.get("transport"), "reason": _short(f.get("reason"), 120)} for f in fb][:4]
        for k in ("preflight", "postflight"):
            if isinstance(full.get(k), dict):
                line[k + "_ok"] = full[k].get("ok")
        cb = full.get("contact_bicgstab")
        if isinstance(cb, dict):
            line["contact_bicgstab"] = {k: (_numbers_only(v) if k == "comm_breakdown" else _short(v, 100) if isinstance(v, str) else v)
                                        for k, v in cb.items()}
    line["detail"] = full.get("detail_file", "bench_detail.json")
    return _num(line)


def _numbers_only(d):
    if not isinstance(d, dict):
        return None
    # (keys without their `_worst_rank` suffix: N > 1 breakdowns are the worst rank's figures throughout; 4 digits)
    out = {k.replace("_worst_rank", ""): _num(v, 4) for k, v in d.items()
           if isinstance(v, (int, float)) and not isinstance(v, bool) and k != "iterations_covered"}
    if "error" in d:
        out["error"] = _short(d["error"], 100)
    if "transport" in d:
        out["transport"] = d["transport"]
    return out


def compact_line(full: dict, cap: int = LINE_CAP) -> str:
    """json.dumps(compact_record(full)) -- never longer than `cap`.  Should a record outgrow the cap all the same, optional blocks
    are dropped in a fixed order (and named in `dropped`) until it fits: the contract's fields, `roofline` and
    `cpu_baseline` always stay."""
    rec = compact_record(full)
    optional = ["blas1_frac", "contact_bicgstab", "comm_breakdown", "configs", "timing", "transport_fallback", "rccl", "transports_measured"]
    dropped = []
    line = json.dumps(rec, separators=(",", ":"), allow_nan=False)
    while len(line) > cap and optional:
        k = optional.pop(0)
        if k in rec:
            del rec[k]
            dropped.append(k)
            rec["dropped"] = dropped
            line = json.dumps(rec, separators=(",", ":"), allow_nan=False)
    if len(line) > cap:  # (cannot happen with the fixed-size blocks that are left; never print a line the driver cannot parse)
        rec["roofline"].pop("note", None)
        if isinstance(rec.get("cpu_baseline"), dict):
            rec["cpu_baseline"].pop("sample", None)
        line = json.dumps(rec, separators=(",", ":"), allow_nan=False)
    assert len(line) <= cap, len(line)
    return line


def emit(full: dict, detail_path) -> None:
    """The full record to `detail_path` (and, as one line, to stderr); the compact line -- the LAST line of stdout."""
    if detail_path:
        full["detail_file"] = os.path.relpath(detail_path, ROOT) if os.path.abspath(detail_path).startswith(ROOT) else detail_path
        try:
            with open(detail_path, "w") as fh:
                json.dump(full, fh, indent=1)
        except OSError as e:
            full["detail_file"] = f"not written ({e.strerror}); see stderr"
    else:
        full["detail_file"] = "stderr only"
    try:
        print("bench_detail " + json.dumps(full), file=sys.stderr, flush=True)
    except Exception:
        pass
    try:
        line = compact_line(full)
    except Exception as e:  # (never lose the run to its own summary: the contract's fields alone)
        print(f"bench.py: compact_line failed ({e!r}); printing the contract's fields only", file=sys.stderr, flush=True)
        r = full.get("roofline") or {}
        c = full.get("cpu_baseline") or {}
        line = json.dumps(_num({
            **{k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                       "scaling", "vs_baseline", "dtype", "data")},
            "config": {"workload": _short((full.get("config") or {}).get("workload"), 200)},
            "roofline": {k: r.get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic")},
            "cpu_baseline": {k: c.get(k) for k in ("value", "unit", "cores", "kind")} if isinstance(c, dict) else None,
            "detail": full.get("detail_file")}), separators=(",", ":"))[:LINE_CAP]
    print(line, flush=True)


class stdout_to_stderr:
    """gloo announces its connections on STDOUT ("[Gloo] Rank 0 is connected to 1 peer ranks...", from C++): while a
    process group is being set up, file descriptor 1 points at stderr, so stdout stays the ONE line the driver parses."""

    def __enter__(self):
        self.saved = None
        try:
            sys.stdout.flush()
            self.saved = os.dup(1)
            os.dup2(2, 1)
        except (OSError, ValueError):  # (no usable descriptor 1 or 2: nothing to protect, nothing to redirect)
            if self.saved is not None:
                os.close(self.saved)
                self.saved = None
        return self

    def __exit__(self, *exc):
        if self.saved is not None:
            try:
                sys.stdout.flush()
            except (OSError, ValueError):
                pass
            os.dup2(self.saved, 1)
            os.close(self.saved)
        return False


def failure_record(args, world, fallbacks) -> dict:
    """What rank 0 prints when NO transport produced a number: the contract's fields with `value` null and what happened."""
    edge = args.n or 256
    return {
        "metric": "CG iterations/sec, 256^3 Poisson per GPU (+ SpMV achieved HBM GB/s in `roofline`)", "value": None,
        "unit": "iter/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": None,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"3D 7-point Poisson {edge}^3 per GPU, fp64 CG [BASELINE.json configs[1]] -- NOT MEASURED",
                   "partition": f"z-slabs, {world} ranks"},
        "roofline": None, "cpu_baseline": None,
        "error": "no transport of the chain produced a result: " + "; ".join(f"{f['transport']}: {f['reason']}" for f in fallbacks),
        "transport_fallback": fallbacks,
    }


RCCL_PLAIN_OPTIONS = ("rccl_flag_wait", "rccl_ticket", "rccl_early_halo", "rccl_fused")


def default_chain(args) -> str:
    """rccl = the transport BASELINE.json's north_star names, with the library's defaults; rccl-plain = the same transport
    with every refinement that has only ever run on a size-1 communicator switched off (RCCL_PLAIN_OPTIONS: cross-stream
    events instead of flag waits, two-launch reductions, no early halo, no fused step) -- run ONLY when rccl did not
    produce a number; ipc = the peer windows; host = staged through host memory, only if nothing else worked."""
    if args.shared_device:
        return "ipc,host"
    return "rccl,rccl-plain" if args.contact else "rccl,rccl-plain,ipc,host"


def contact_defaults(args) -> None:
    """--contact: a short lease on N devices must still yield a parsed record.  Small blocks, no side measurements."""
    if args.n is None:
        args.n = 64
    args.skip_general = args.skip_blas1 = args.skip_permuted = args.skip_unstructured = True
    args.skip_unstructured3d = args.skip_spmv = args.skip_configs = True
    args.traffic, args.cpu_iters = "off", 0
    args.spinup_seconds = min(args.spinup_seconds, 0.2)
    args.min_seconds = min(args.min_seconds, 0.05)
    args.roofline_launches = min(args.roofline_launches, 50)
    if args.attempt_seconds == "240,150,150":
        args.attempt_seconds = "90,90,60"


def baseline_configs(api, mesh, ctx, op, b, N, n, st, min_seconds, self_exchange=("cg", "bicgstab")):
    """BASELINE configs 3, 4 and 5 on one GPU, each bounded to a few hundred milliseconds of device time:
      3  BiCGStab on the headline 256^3 block (the per-GPU problem of the 8-GPU config; SolverBiCgStab.hpp:93-165);
      4  GMRES(30) on the 128^3 convection-diffusion operator (SolverGmres.hpp:119-249);
      5  the lid-driven cavity's time step at 128^3 (pressure-Poisson CG each step; Playground.cpp:186-206's loop shape).
    Fixed iteration counts with the tolerances off where a rate is quoted; bytes are those the kernels really move."""
    import numpy as np

    from stormruler_amd import cavity

    out = {}

    def rate(cls, operator, rhs, rows, iters, setup=None):
        reps = []
        # (one untimed solve first: a context's first solve of a kind allocates its work vectors -- GMRES's basis is a
        #  6 GB arena at 256^3 -- and loads the kernels' code objects)
        while len(reps) < 2 or (sum(reps[1:]) < min_seconds and len(reps) < 51):
            s_ = cls()
            if setup:
                setup(s_)
            s_.num_iterations, s_.absolute_error_tolerance, s_.relative_error_tolerance = iters, 0.0, 0.0
            x_ = api.DeviceVector(ctx, rows)
            ctx.sync()
            t0 = time.perf_counter()
            s_.solve(x_, rhs, operator)
            ctx.sync()
            reps.append(time.perf_counter() - t0)
        return float(np.median(reps[1:])) / iters, len(reps) - 1

    # ---- config 1 (the reference's own CPU-runnable case: CG on the 64^3 box; the CPU side of it is in cpu_baseline's sample)
    try:
        g1 = mesh.structured_box(64)
        m1 = api.StencilMatrix.from_face_graph(ctx, g1)
        b1 = api.DeviceVector(ctx, g1.n_cells)
        api.fill_with(b1, 1.0)
        r0 = ctx.counter("resident_solves")
        sec, reps = rate(api.CgSolver, api.HipStencilOperator(m1, -1.0, 0.0), b1, g1.n_cells, 2000)
        out["config1_cg64"] = {
            "workload": "CG, 64^3 Poisson (BASELINE configs[0]), 2 000 iterations, tolerances off",
            "iter_per_s": 1.0 / sec, "us_per_iteration": sec * 1e6,
            "path": "resident (one persistent kernel per solve)" if ctx.counter("resident_solves") > r0 else "other", "repeats": reps}
        m1.close()
    except Exception as e:
        out["config1_cg64"] = {"error": repr(e)}
    # ---- config 3
    try:
        sec, reps = rate(api.BiCgStabSolver, op, b, N, 60)
        # what the five kernels of an iteration stream: two applies (records + x + y), r~ beside the first one for <r~, v>,
        # p = r + beta (p - omega v) 32 N, s = r - alpha v 24 N, the second half-step 56 N (x, p, s, t, r~ read; x, r written:
        # seven streams) = records x 2 + 152 N = 168 B/row on the lattice records.  (Until round 6 this line counted
        # 104 N for the vector passes -- NOTES.md section 5b's figure, which leaves r~'s two reads to the dots -- and
        # under-reported `frac` by a tenth.)
        moved = 2 * (st["record_bytes"] + 16 * N) + 8 * N + 32 * N + 24 * N + 56 * N
        out["config3_bicgstab256"] = {
            "workload": f"BiCGStab, {n}^3 Poisson block (BASELINE configs[2]'s per-GPU problem), 60 iterations, tolerances off",
            "iter_per_s": 1.0 / sec, "us_per_iteration": sec * 1e6, "bytes_really_moved_per_iteration": moved,
            "frac": moved / sec / 1e9 / HBM_PEAK_GBS, "reference_op_list_bytes_per_iteration": 2 * (24 * N + 12 * st["nnz_offdiag"]) + 192 * N,
            "repeats": reps}
    except Exception as e:
        out["config3_bicgstab256"] = {"error": repr(e)}
    # ---- (not a BASELINE config: the kernel-per-statement GMRES at the headline size -- the Gram-Schmidt passes of
    #       csrc/solvers.hip, mgs_multi_kernel, at HBM scale)
    try:
        def m30h(s_):
            s_.num_inner_iterations = 30

        sec, reps = rate(api.GmresSolver, op, b, N, 60, m30h)
        # inner iteration k with four steps per pass: the apply, (k + 1) basis vectors read twice (projected on, subtracted),
        # w read and written once per pass (ceil((k + 1) / 4) + 1 passes), the normalised q_{k+1} written; mean over k
        passes = float(np.mean([-(-(k + 1) // 4) + 1 for k in range(30)]))
        moved = (st["record_bytes"] + 16 * N) + 16 * N * 15.5 + 16 * N * passes + 16 * N
        out["extra_gmres30_poisson256"] = {
            "workload": f"GMRES(30), {n}^3 Poisson block, 60 inner iterations (two restarts), tolerances off",
            "iter_per_s": 1.0 / sec, "us_per_inner_iteration": sec * 1e6, "bytes_really_moved_per_inner_iteration_mean": moved,
            "frac": moved / sec / 1e9 / HBM_PEAK_GBS,
            "reference_mgs_bytes_per_inner_iteration_mean": (24 * N + 12 * st["nnz_offdiag"]) + 15.5 * 40 * N + 24 * N, "repeats": reps}
    except Exception as e:
        out["extra_gmres30_poisson256"] = {"error": repr(e)}
    # ---- config 4
    try:
        g4 = mesh.structured_box(128)
        wi, wo, de = mesh.convection_diffusion_weights(g4, 1e-2, (1.0, 0.5, 0.25))
        m4 = api.StencilMatrix.from_face_weights(ctx, g4.n_cells, g4.n_halo, g4.inner, g4.outer, wi, wo, de)
        st4 = m4.stats()
        b4 = api.DeviceVector(ctx, g4.n_cells)
        api.fill_with(b4, 1.0)
        op4 = api.HipStencilOperator(m4, 1.0, 0.0)

        def m30(s_):
            s_.num_inner_iterations = 30

        sec, reps = rate(api.GmresSolver, op4, b4, g4.n_cells, 600, m30)
        n4 = g4.n_cells
        # inner iteration k, ONE chain launch: the apply's records and its gathers of q_k, every basis vector q_0 .. q_k once
        # (8 B/row each; mean over k = 0 .. 29: 15.5), q_{k+1} out -- w never leaves the registers; per cycle of 30 the
        # restart: x += sum beta_i q_i (30 vectors, x in and out once per launch of eight) and r = b - A x, its norm, q_0
        moved = (st4["record_bytes"] + 8 * n4) + 8 * n4 * 15.5 + 8 * n4 + (30 * 8 * n4 + 4 * 16 * n4 + (st4["record_bytes"] + 16 * n4 + 24 * n4 + 8 * n4 + 16 * n4)) / 30.0
        out["config4_gmres30_convdiff128"] = {
            "workload": "GMRES(30), 128^3 convection-diffusion (nu = 1e-2, v = (1, 0.5, 0.25), first-order upwind), 600 "
                        "inner iterations, tolerances off [BASELINE configs[3]]",
            "iter_per_s": 1.0 / sec, "us_per_inner_iteration": sec * 1e6, "record_format": record_format_name(st4),
            "fused_bytes_per_inner_iteration_mean": moved, "frac": moved / sec / 1e9 / HBM_PEAK_GBS,
            "reference_mgs_bytes_per_inner_iteration_mean": (24 * n4 + 12 * st4["nnz_offdiag"]) + 15.5 * 40 * n4 + 24 * n4,
            "repeats": reps}
        m4.close()
    except Exception as e:
        out["config4_gmres30_convdiff128"] = {"error": repr(e)}
    # ---- config 5
    try:
        t0 = time.perf_counter()
        dev = cavity.CavityProjection(ctx, 128, nu=0.01)
        t_setup = time.perf_counter() - t0
        its, secs = [], []
        r0 = ctx.counter("resident_solves")
        for _ in range(6):
            it_, sec_, ok_ = dev.step()
            its.append(int(it_)), secs.append(float(sec_))
        out["config5_cavity128"] = {
            "workload": "lid-driven cavity 128^3 (Chorin projection: 18 SpMVs + one warm-started pressure-Poisson CG per "
                        "step), 6 steps from rest, one GPU [BASELINE configs[4]'s per-problem size]",
            "s_per_step": float(np.median(secs[2:])), "s_per_step_all": secs, "cg_iterations_per_step": its,
            "us_per_cg_iteration_upper_bound": 1e6 * float(np.median(secs[2:])) / max(int(np.median(its[2:])), 1),
            "pressure_solves_on_the_resident_path": ctx.counter("resident_solves") - r0, "setup_seconds": t_setup}
        del dev
    except Exception as e:
        out["config5_cavity128"] = {"error": repr(e)}
    # ---- a USER's host loop: the reference's CG body typed statement by statement against Storm.hpp (the C++ driver's
    # `user-cg`, tests/cpp/poisson_driver.cpp; SolverCg.hpp:86-126) with the library's lazy statements, with every statement a
    # launch of its own, and the library's device loop -- child processes of this one, one after the other ----
    try:
        out["host_loop_cg256"] = host_loop_rates(n)
    except Exception as e:
        out["host_loop_cg256"] = {"error": repr(e)}
    # ---- what the multi-rank code path costs with the network taken out: ONE rank on a size-1 RCCL communicator, a
    # z-periodic box whose two halo planes it exchanges with itself (tools/comm_path_overhead.py, child processes) ----
    try:
        if self_exchange:
            out["multi_rank_path_at_one_rank"] = self_exchange_rates(n, out, self_exchange)
    except Exception as e:
        out["multi_rank_path_at_one_rank"] = {"error": repr(e)}
    return out


def self_exchange_rates(n, configs, solvers=("cg", "bicgstab")):
    import subprocess

    tool = os.path.join(ROOT, "tools", "comm_path_overhead.py")
    o = {"workload": f"{n}^3 z-periodic box on ONE rank, communicator of size 1 (RCCL): both halo planes exchanged with the rank "
                     "itself, every reduction through the all-reduce; 400 iterations, tolerances off; "
                     "`COMM_SOLVER=... COMM_ITERS=400 python tools/comm_path_overhead.py <n> rccl`"}
    for solver in solvers:
        p = subprocess.run([sys.executable, tool, str(n), "rccl"], capture_output=True, text=True, timeout=600, cwd=ROOT,
                           env=dict(os.environ, COMM_SOLVER=solver, COMM_ITERS="400"))
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        if p.returncode != 0 or not lines:
            raise RuntimeError((p.stdout + p.stderr)[-500:])
        d = json.loads(lines[-1])
        br = d.get("rccl_mixed_comm_breakdown") or {}
        o[solver] = {"us_per_iteration_over_rccl": 1e6 / d["rccl_mixed_it_per_s"],
                     "comm_breakdown": {k: v for k, v in br.items() if k != "note"}}
    plain_bicg = (configs.get("config3_bicgstab256") or {}).get("us_per_iteration")
    if plain_bicg and "bicgstab" in o:
        o["bicgstab"]["us_per_iteration_plain"] = plain_bicg
        o["bicgstab"]["overhead_us_per_iteration"] = o["bicgstab"]["us_per_iteration_over_rccl"] - plain_bicg
    return o


def host_loop_rates(n, iterations=200):
    import subprocess

    driver = os.path.join(ROOT, "tests", "cpp", "poisson_driver")
    if not os.path.exists(driver):
        return {"error": "tests/cpp/poisson_driver is not built (__graft_entry__.build() builds it)"}

    def run(kind, mode):
        p = subprocess.run([driver, str(n), kind, mode], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, DRIVER_FIXED_ITERATIONS=str(iterations)))
        if p.returncode != 0:
            raise RuntimeError((p.stdout + p.stderr)[-500:])
        for ln in p.stdout.splitlines():
            if "timed_solve_seconds" in ln:
                d = json.loads(ln)
                return d["timed_solve_seconds"] / d["timed_iterations"] * 1e6
        raise RuntimeError(p.stdout[-500:])

    o = {"workload": f"CG, {n}^3 Poisson, {iterations} iterations, tolerances off; `tests/cpp/poisson_driver {n} user-cg|cg native|eager`",
         "device_loop_us_per_iteration": run("cg", "native"),
         "host_loop_lazy_statements_us_per_iteration": run("user-cg", "native"),
         "host_loop_eager_statements_us_per_iteration": run("user-cg", "eager")}
    o["host_loop_lazy_over_device_loop"] = o["host_loop_lazy_statements_us_per_iteration"] / o["device_loop_us_per_iteration"]
    return o


def spmv_standalone(api, ctx, matrix, stats, cell_ids, launches, traffic_bytes=None):
    """The "SpMV achieved HBM GB/s" half of BASELINE.json's metric as SURVEY.md 8d specifies it: stand-alone
    `storm_hip_op_apply` launches (y = -L x, nothing fused in) on x_i = sin(0.37 i), i = global cell id, after a
    warm-up; one HIP-event pair per launch on the library's compute stream; MEDIAN and mean.  Measured twice:
    `back_to_back` on one (x, y) pair -- part of whose 2 x 8 B/row may be served by the 256 MiB Infinity Cache from
    one launch to the next -- and `rotating` over three pairs (6 vectors: > 800 MB at 256^3, every launch finds its x
    and y evicted).  Each time is priced three ways: SURVEY 8d's algorithmic bytes (24 N + 12 nnz: fp64 weights +
    int32 columns, may exceed the peak for the byte-indexed formats -- those do not move these bytes), the bytes the
    record format really streams (records + x + y), and the PMC-measured HBM traffic of the same kernel."""
    import numpy as np

    n_rows = stats["n_rows"]
    alg = 24 * n_rows + 12 * stats["nnz_offdiag"]
    streamed = stats["record_bytes"] + 16 * n_rows
    xh = np.sin(0.37 * np.asarray(cell_ids, dtype=np.float64))
    n_h = stats["n_cols"] - n_rows
    xs_ = [api.DeviceVector(ctx, n_rows, n_h) for _ in range(3)]
    ys_ = [api.DeviceVector(ctx, n_rows, n_h) for _ in range(3)]
    for v_ in xs_:
        v_.upload(xh)
    ctx.set_option("profile_spmv", 1)
    for i_ in range(12):  # warm-up (timed only for `all_launches_mean_ms`, the figure a rocprofv3 --stats average of a
        matrix.apply(-1.0, 0.0, xs_[i_ % 3], ys_[i_ % 3])  # --spmv-only run compares with)
    all_ms = [ctx.spmv_profile_samples()]
    ctx.set_option("profile_spmv", 0)

    def timed(pairs):
        ctx.set_option("profile_spmv", 1)
        for i_ in range(launches):
            matrix.apply(-1.0, 0.0, xs_[i_ % pairs], ys_[i_ % pairs])
        smp = ctx.spmv_profile_samples()
        ctx.set_option("profile_spmv", 0)
        all_ms.append(smp)
        per_apply = smp.size // launches  # (a partitioned operator: interior + boundary launch per apply)
        t_ = smp[: per_apply * launches].reshape(launches, per_apply).sum(axis=1)
        med, mean = float(np.median(t_)), float(t_.mean())

        def gbs(b_, ms_):
            return b_ / (ms_ * 1e-3) / 1e9

        o_ = {"launches": launches, "median_ms": med, "mean_ms": mean, "min_ms": float(t_.min()), "max_ms": float(t_.max()),
              "GBs_8d_bytes": gbs(alg, med), "frac_8d": gbs(alg, med) / HBM_PEAK_GBS,
              "GBs_streamed_bytes": gbs(streamed, med), "frac_streamed": gbs(streamed, med) / HBM_PEAK_GBS,
              "GBs_8d_bytes_by_mean": gbs(alg, mean), "GBs_streamed_bytes_by_mean": gbs(streamed, mean)}
        if traffic_bytes:
            o_["GBs_pmc_traffic"] = gbs(traffic_bytes, med)
            o_["frac_pmc_traffic"] = gbs(traffic_bytes, med) / HBM_PEAK_GBS
        return o_

    out_ = {"kernel": kernel_name(stats).split(" / ")[-1], "record_format": record_format_name(stats),
            "input": "x_i = sin(0.37 i), i = global cell id; y = -L x (alpha = -1, beta = 0); storm_hip_op_apply alone",
            "rows": n_rows, "nnz_offdiag": stats["nnz_offdiag"], "algorithmic_bytes_8d": alg, "streamed_bytes": streamed,
            "pmc_traffic_bytes": traffic_bytes, "ell_padding_ratio": stats["ell_slots"] / max(stats["nnz_offdiag"] - stats["tail_nnz"], 1) - 1.0,
            "tail_nnz": stats["tail_nnz"],
            "back_to_back": timed(1), "rotating_3_pairs": timed(3)}
    # the y of the last launch against a second evaluation order is not a parity check (tests/ hold those): only
    # that the launches did something -- |y| is finite and non-zero
    out_["y_norm"] = float(api.norm_2(ys_[0]))
    # What an event pair reads for a kernel that does (almost) nothing: the same instrumented launch of a 64-row operator.
    # rocprofv3's kernel durations do not contain it -- its averages for these kernels are this much shorter (4 - 5 us).
    # (In a context of its own, with the default options: a dictionary-format operator -- `spmv_canon_kernel`, a name none of
    #  the measured launches has; 45 launches of 2 us under the measured kernel's name would drag rocprofv3's average for
    #  it down by a quarter.)
    try:
        from stormruler_amd import mesh as _mesh

        fctx = api.Context(0)
        tiny = api.StencilMatrix.from_face_graph(fctx, _mesh.structured_box(4))
        tx, ty = api.DeviceVector(fctx, 64), api.DeviceVector(fctx, 64)
        for _ in range(5):
            tiny.apply(-1.0, 0.0, tx, ty)
        fctx.set_option("profile_spmv", 1)
        for _ in range(40):
            tiny.apply(-1.0, 0.0, tx, ty)
        floor = fctx.spmv_profile_samples()
        fctx.set_option("profile_spmv", 0)
        out_["hip_event_pair_floor_ms"] = float(np.median(floor))
        out_["hip_event_pair_floor_kernel"] = kernel_name(tiny.stats()).split(" / ")[-1]
        del tx, ty
        tiny.close()
        fctx.close()
    except Exception as e_:  # noqa: BLE001
        out_["hip_event_pair_floor_ms"] = None
        out_["hip_event_pair_floor_error"] = repr(e_)[:200]
    cat = np.concatenate(all_ms)
    out_["all_launches"] = int(cat.size)
    out_["all_launches_mean_ms"] = float(cat.mean())
    return out_


def spmv_only(args) -> int:
    """`--spmv-only`: nothing but the `spmv` block (lattice records, then fp64 records) in a process of its own, so that a
    `rocprofv3 --kernel-trace --stats` of this command holds exactly these launches per kernel: its AverageNs for
    spmv_canon_tile_kernel<false, ...> / spmv_sell_kernel<true, false, ...> is `all_launches_mean_ms`."""
    import numpy as np

    from stormruler_amd import api, mesh

    g = mesh.structured_box(args.n)
    ctx = api.Context(0)
    for kv in args.opt:
        k_, v_ = kv.split("=")
        ctx.set_option(k_, int(v_))
    out = {}
    what = args.spmv_what.split(",")
    for name, level in (("lattice", None), ("general", 0)):
        if name not in what:
            continue
        if level is not None:
            ctx.set_option("spmv_dict", level)
        mat = api.StencilMatrix.from_face_graph(ctx, g)
        out[name] = spmv_standalone(api, ctx, mat, mat.stats(), np.arange(g.n_cells), args.spmv_launches)
        mat.close()
    if "tets" in what:  # the tetrahedral mesh of roofline_unstructured3d (a run of its own: the kernel is `general`'s)
        import shutil
        import tempfile

        d = tempfile.mkdtemp(prefix="storm_tet_", dir=os.environ.get("TMPDIR", "/tmp"))
        try:
            prefix, _, _ = tet_files(args.tet_edge, d)
            hm, mat, _ = tet_operator(api, ctx, prefix)
            st = mat.stats()
            gid = np.ctypeslib.as_array(hm.view().global_id, shape=(st["n_rows"],)).copy()
            out["tets"] = spmv_standalone(api, ctx, mat, st, gid, args.spmv_launches)
            mat.close()
            hm.close()
        finally:
            shutil.rmtree(d, ignore_errors=True)
    print(json.dumps({"spmv": out, "n": args.n, "device": ctx.info()["name"]}), flush=True)
    ctx.close()
    return 0



def tet_files(n3, workdir):
    """The seeded tetrahedral box (stormruler_amd.io_tetgen.tet_box: 6 n3^3 cells) written as TetGen files by the library's
    writer.  Returns (prefix, seconds by step, bytes on disk)."""
    import numpy as np

    from stormruler_amd import host_mesh, io_tetgen

    sec = {}
    t0 = time.time()
    pos, bf, cells = io_tetgen.tet_box(n3)
    sec["generate_numpy"] = time.time() - t0
    prefix = os.path.join(workdir, "tetbox.1")
    t0 = time.time()
    host_mesh.write_tetgen(prefix, pos, bf, np.ones(len(bf), np.int64), cells)
    sec["write_tetgen_files"] = time.time() - t0
    size = sum(os.path.getsize(prefix + e) for e in (".node", ".edge", ".face", ".ele"))
    return prefix, sec, size


def tet_operator(api, ctx, prefix, file_order_probe=None):
    """<prefix>.node/.edge/.face/.ele -> the library's reader (3-D branch of read_mesh_from_tetgen) -> Morton order of the
    cell centres -> the operator.  Returns (host mesh, operator, seconds by step)."""
    from stormruler_amd import host_mesh

    sec = {}
    t0 = time.time()
    hm = host_mesh.HostMesh.read_tetgen(prefix + ".", 3)
    sec["read_files_and_build_face_graph"] = time.time() - t0
    if file_order_probe is not None:  # the operator in FILE order first: what the ordering must conjugate, not change
        m0 = hm.create_operator(ctx)
        file_order_probe(m0)
        m0.close()
    t0 = time.time()
    kind = hm.order_cells("morton")
    sec["morton_ordering"] = time.time() - t0
    assert kind == "morton"
    t0 = time.time()
    mat = hm.create_operator(ctx)
    sec["operator_build"] = time.time() - t0
    return hm, mat, sec


def tet_variant(api, ctx, args, prefix, file_seconds, file_bytes, traffic_bytes, traffic_dot_bytes):
    """`roofline_unstructured3d`: a genuinely unstructured 3-D mesh at HBM scale -- tetrahedra read from TetGen files,
    rows of 2 - 4 neighbours with all-distinct fp64 weights, Z-order numbering -- through the kernel every mesh gets
    (spmv_sell_kernel, fp64 records: streamed bytes == SURVEY 8d's algorithmic bytes, but for the ELL padding)."""
    import numpy as np

    probe = {}

    def file_order_probe(m0):  # K CG iterations on the operator as the files number it
        n0 = m0.stats()["n_rows"]
        b0, x0 = api.DeviceVector(ctx, n0), api.DeviceVector(ctx, n0)
        api.fill_with(b0, 1.0)
        s0 = api.CgSolver()
        s0.num_iterations, s0.absolute_error_tolerance, s0.relative_error_tolerance = 20, 0.0, 0.0
        s0.solve(x0, b0, api.HipStencilOperator(m0, -1.0, 0.0))
        probe["residual"] = s0.absolute_error

    hm, mat, sec = tet_operator(api, ctx, prefix, file_order_probe)
    sec = dict(file_seconds, **sec)
    st = mat.stats()
    v = hm.view()
    n_rows = st["n_rows"]
    gid = np.ctypeslib.as_array(v.global_id, shape=(n_rows,)).copy()  # row i is cell gid[i] of the files
    op = api.HipStencilOperator(mat, alpha=-1.0, beta=0.0)
    b = api.DeviceVector(ctx, n_rows)
    api.fill_with(b, 1.0)
    K = args.steps

    def run(iters):
        x = api.DeviceVector(ctx, n_rows)
        s_ = api.CgSolver()
        s_.num_iterations, s_.absolute_error_tolerance, s_.relative_error_tolerance = iters, 0.0, 0.0
        s_.solve(x, b, op)
        assert s_.iteration == iters
        return s_

    run(max(args.warmup, 20))
    ctx.sync()
    reps = []
    while sum(reps) < args.min_seconds and len(reps) < 2000:
        t1 = time.perf_counter()
        s_ = run(K)
        ctx.sync()
        reps.append(time.perf_counter() - t1)
    t1 = float(np.median(reps))
    # the SpMV launches of a CG solve (fused <p, Ap> partials), HIP-event pairs
    iters = max(K, args.roofline_launches)
    ctx.set_option("profile_spmv", 1)
    run(iters)
    smp = ctx.spmv_profile_samples()
    ctx.set_option("profile_spmv", 0)
    alg = 24 * n_rows + 12 * st["nnz_offdiag"]
    streamed = st["record_bytes"] + 16 * n_rows
    ms, med = float(smp.mean()), float(np.median(smp))
    out = {"kernel": "spmv_sell_kernel (fp64 records) + fused <p,Ap> partials", "record_format": record_format_name(st),
           "mesh": f"tetrahedral box, {st['n_rows']} cells (6 x {args.tet_edge}^3: Kuhn's six tetrahedra per cube, interior nodes "
                   "jittered by <= 0.15 h, seeded), written as TetGen .node/.edge/.face/.ele by storm_hip_mesh_write_tetgen, read back by "
                   "storm_hip_mesh_read_tetgen (3-D branch of Mallard/IoTetgen.hpp), cells renumbered along the Z-order curve of "
                   "their centres (storm_hip_order_cells), operator by storm_hip_op_create_from_mesh_object",
           "bound": "hbm", "achieved": alg / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "frac_8d": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "frac_by_median": alg / (med * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "frac_streamed_bytes": streamed / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
           "traffic": traffic_dot_bytes, "traffic_over_8d_bytes": (traffic_dot_bytes / alg) if traffic_dot_bytes else None,
           "algorithmic_bytes_8d": alg, "streamed_bytes": streamed, "avg_launch_ms": ms, "median_launch_ms": med,
           "min_launch_ms": float(smp.min()), "launches_timed": int(smp.size),
           "rows": n_rows, "interior_faces": int(v.n_faces), "boundary_faces": int(v.n_bfaces), "nnz_offdiag": st["nnz_offdiag"],
           "max_row_len": st["max_row_len"], "ell_slots": st["ell_slots"],
           "ell_padding_ratio": st["ell_slots"] / max(st["nnz_offdiag"] - st["tail_nnz"], 1) - 1.0, "tail_nnz": st["tail_nnz"],
           "cg_iter_per_s": K / t1, "ms_per_step": t1 / K * 1e3, "cg_final_residual": s_.absolute_error,
           # b = 1 is invariant under a renumbering: the ordered operator is the file-order one conjugated by a permutation
           # (the SpMV of the two agrees BIT FOR BIT at this size: tools/tet_conjugation_check.py), so 20 CG iterations leave
           # the same residual to the rounding of differently grouped sums.  (Not 200: on this operator -- rows divided by
           # their cell volumes, so not symmetric, as the reference's -- CG amplifies a last-place difference tenfold every
           # ~20 iterations: 1e-16 at iteration 10, 6e-9 at 50, 5e-2 at 200; profiles/r08g_tet_conjugation.json.)
           "residual_after_20_iterations_rel_diff_vs_file_order": abs(run(20).absolute_error - probe["residual"]) / probe["residual"] if probe else None,
           "cg_reference_op_list_bytes_per_iteration": alg + 96 * n_rows,
           "host_seconds": sec, "tetgen_files_bytes": file_bytes}
    band = np.abs(np.ctypeslib.as_array(v.inner, shape=(v.n_faces,)) - np.ctypeslib.as_array(v.outer, shape=(v.n_faces,)))
    out["column_distance_median"], out["column_distance_p99"], out["column_distance_max"] = (
        int(np.median(band)), int(np.percentile(band, 99)), int(band.max()))
    if not args.skip_spmv:
        out["spmv"] = spmv_standalone(api, ctx, mat, st, gid, args.spmv_launches, traffic_bytes)
    mat.close()
    hm.close()
    return out


def record_format_name(st) -> str:
    return ("canonical paired rows: byte-indexed weights, one common offset order (8 B/row)" if st["paired_rows"] == 2 else
            "paired rows: byte-indexed weights + shared column offsets (12 B/row)" if st["paired_rows"] else
            "byte-indexed weights + column offsets (16 B/row)" if st["offset_dictionary_size"] else
            "byte-indexed weights (8 B/row + int32 columns)" if st["value_dictionary_size"] else
            "fp64 weights + int32 columns")


def kernel_name(st) -> str:
    return ("cg_step_march_kernel (fused CG step) / spmv_canon_tile_kernel" if st["paired_rows"] >= 2 and st.get("tiled_planes") else
            "spmv_canon_kernel" if st["paired_rows"] >= 2 else "spmv_pair_kernel" if st["paired_rows"] else
            "spmv_dict_kernel" if st["value_dictionary_size"] and st.get("uniform_width", 1) else
            "spmv_sell_kernel")


def run_preflight(api, dist, mesh, partition, connect, local_rank, world, rank, wrong=False):
    """Bounded pre-flight of a transport, before the 256^3 setup: a 16 x 16 x (8 world) slab stack, ONE halo-exchanged
    SpMV and ONE all-reduce, both checked numerically.  x = the global z index of a cell: rows away from the walls
    must give exactly 0 (every difference x_nb - x_i is +-1 or 0 and the weights cancel in pairs), which fails at the
    slab interfaces iff a halo plane is stale or misplaced; <1, 1> must be the global cell count exactly."""
    import numpy as np

    t0 = time.perf_counter()
    nx = ny = 16
    nzl = 8
    g, plan = partition.slab_partition(nx, ny, nzl, world, rank)
    ctx = api.Context(local_rank)
    connect(ctx)
    mat = api.StencilMatrix.from_face_graph(ctx, g)
    if plan.n_nbrs:
        mat.set_halo(plan.nbr_rank, plan.send_ptr, plan.send_idx, plan.recv_ptr)
    op = api.HipStencilOperator(mat, alpha=1.0, beta=0.0)
    gid = np.asarray(g.global_id[:g.n_cells], dtype=np.int64)
    kz = (gid // (nx * ny)).astype(np.float64)
    x = api.DeviceVector(ctx, g.n_cells, g.n_halo)
    x.upload(kz + (1.0 if wrong else 0.0) * (rank == 1))
    y = api.DeviceVector(ctx, g.n_cells, g.n_halo)
    op.mul(y, x)
    yv = y.to_numpy()
    i, j = gid % nx, (gid // nx) % ny
    kg = gid // (nx * ny)
    inner = (i > 0) & (i < nx - 1) & (j > 0) & (j < ny - 1) & (kg > 0) & (kg < nzl * world - 1)
    bad = int(np.count_nonzero(yv[inner] != 0.0))
    ones = api.DeviceVector(ctx, g.n_cells, g.n_halo)
    api.fill_with(ones, 1.0)
    total = api.dot_product(ones, ones)
    ctx.sync()
    ok = bad == 0 and total == float(nx * ny * nzl * world)
    worst = dist.allreduce_max(0.0 if ok else 1.0)
    mat.close()
    dist.barrier()
    ctx.close()
    res = {"ok": worst == 0.0, "halo_rows_wrong_on_this_rank": bad, "allreduce_sum": total,
           "expected_sum": float(nx * ny * nzl * world), "seconds": time.perf_counter() - t0}
    if worst != 0.0:
        print(f"bench.py: pre-flight FAILED on rank {rank}: {res}", file=sys.stderr, flush=True)
        os._exit(42)
    return res


def supervise(args, chain) -> int:
    """One rank of an N > 1 run as torch.distributed.run (the driver's, or launch_ranks' below) starts it.  This process
    never touches the GPU: it starts the measuring rank as a CHILD per transport attempt, watches it against a
    wall-clock budget together with the other ranks' supervisors (gloo, once a second), and on any failure kills the
    children and starts a fresh set on the next transport.  Rank 0 relays the JSON line, adding what happened."""
    import signal
    import socket
    import subprocess

    import torch
    import torch.distributed as td

    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")

    def cpu_barrier():
        # (NOT td.barrier(): that one asks torch for "the accelerator on the machine" first, which opens the GPU -- and a
        #  supervisor never does: N of them beside N measuring ranks would double the processes that hold the device)
        td.all_reduce(torch.zeros(1, dtype=torch.float64))

    with stdout_to_stderr():
        td.init_process_group("gloo", rank=rank, world_size=world)
        cpu_barrier()  # (the connections are made -- and announced -- at the first collective)
    budgets = [float(v) for v in args.attempt_seconds.split(",")]
    fallbacks, line = [], None
    measured, n_measured, measured_names = {}, 0, []  # transport -> parsed record (rank 0 holds them, every rank the names)
    for attempt, transport in enumerate(chain):
        if n_measured and (transport == "host" or args.one_transport):
            break  # host-staged is a fallback only; --one-transport: the first transport that works
        if transport == "rccl-plain" and "rccl" in measured_names:
            continue  # the plain form of the RCCL transport: only when the default form gave no number
        budget = budgets[min(attempt, len(budgets) - 1)]
        port = torch.zeros(1, dtype=torch.int64)
        if rank == 0:
            with socket.socket() as sock:
                sock.bind(("127.0.0.1", 0))
                port[0] = sock.getsockname()[1]
        td.broadcast(port, src=0)
        env = dict(os.environ, STORM_BENCH_WORKER="1", STORM_BENCH_TRANSPORT=transport, MASTER_PORT=str(int(port[0])),
                   STORM_BENCH_PREFLIGHT_SECONDS=str(int(min(120, budget))))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL's own words when something goes wrong (bootstrap, peer access, a refused split): warnings only, to stderr
        # (the child's stdout is the record the supervisor parses)
        env.setdefault("NCCL_DEBUG", "WARN")
        env.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_USE_AGENT_STORE"):  # the child makes its own rendezvous on `port`
            env.pop(k, None)
        out_path = os.path.join(os.environ.get("TMPDIR", "/tmp"), f"storm_bench_{os.getpid()}_{attempt}.out")
        with open(out_path, "w") as fo:
            child = subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env, stdout=fo,
                                     stderr=None, start_new_session=True)
        t0 = time.perf_counter()
        reason = None
        while True:
            rc = child.poll()
            mine = torch.tensor([1.0 if rc == 0 else 0.0, 1.0 if (rc is not None and rc != 0) else 0.0,
                                 1.0 if (rank == 0 and time.perf_counter() - t0 > budget) else 0.0], dtype=torch.float64)
            td.all_reduce(mine)  # sums: ranks done ok, ranks failed, rank 0's clock says "over budget"
            if mine[1] > 0:
                reason = f"a rank's process failed (this rank's exit code: {rc})"
                break
            if mine[2] > 0:
                reason = f"no result within the budget of {budget:.0f} s"
                break
            if mine[0] == world:
                break
            time.sleep(1.0)
        if reason is None:
            text = open(out_path).read()
            lines = [ln for ln in text.splitlines() if ln.startswith("{")]
            line = lines[-1] if lines else None
            ok = torch.tensor([1.0 if (rank != 0 or line is not None) else 0.0], dtype=torch.float64)
            td.all_reduce(ok, op=td.ReduceOp.MIN)
            if ok[0] == 0:
                reason = "rank 0 printed no result line"
        try:
            os.remove(out_path)
        except OSError:
            pass
        if reason is None:
            n_measured += 1
            measured_names.append(transport)
            if rank == 0:
                measured[transport] = json.loads(line)
            cpu_barrier()
            continue  # ... on to the next transport of the chain: both RCCL and the peer windows are measured
        if child.poll() is None:  # end exactly the process group this supervisor started
            try:
                os.killpg(child.pid, signal.SIGKILL)
            except ProcessLookupError:
                pass
        child.wait()
        fallbacks.append({"transport": transport, "reason": reason, "seconds": time.perf_counter() - t0})
        if rank == 0:
            print(f"bench.py: transport {transport}: {reason}; " +
                  (f"starting fresh ranks on {chain[attempt + 1]}" if attempt + 1 < len(chain) else "no transport left"),
                  file=sys.stderr, flush=True)
        line = None
        cpu_barrier()
    status = 0 if n_measured else 1
    if rank == 0 and measured:
        best = max(measured, key=lambda t: measured[t]["value"])
        out = measured[best]
        out["transports_measured"] = {
            t: {k: m.get(k) for k in ("value", "ms_per_step", "timing", "comm_breakdown", "postflight")} for t, m in measured.items()}
        out["transport_choice"] = ("`value` / `ms_per_step` are the better of the transports measured in this run (each by fresh rank "
                                   "processes: rccl = the transport BASELINE.json's north_star names, ipc = the library's "
                                   "peer windows); `transports_measured` holds every one")
        out["transport_fallback"] = fallbacks
        emit(out, args.detail_path)
    elif rank == 0:  # nothing measured: still ONE line, saying so (the exit code stays non-zero)
        emit(failure_record(args, world, fallbacks), args.detail_path)
    flag = torch.tensor([float(status)], dtype=torch.float64)
    td.all_reduce(flag, op=td.ReduceOp.MAX)
    td.destroy_process_group()
    return int(flag[0])


def launch_ranks(n_ranks: int, args) -> int:
    """`python bench.py --gpus N` outside a launcher: start the N rank supervisors as CHILD processes (one per GPU,
    torch.distributed.run on 127.0.0.1) with this command line, relay their output (rank 0 prints the JSON
    line) and return their exit code.  Nothing in this process has touched torch or HIP; no exec.  Bounded: the
    launcher and everything below it is ended when the attempts' budgets (plus slack) are spent."""
    import signal
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__), *sys.argv[1:]]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    n_chain = len((args.transport or default_chain(args)).split(","))
    budgets = [float(v) for v in args.attempt_seconds.split(",")]
    total = sum(budgets[min(i, len(budgets) - 1)] for i in range(n_chain)) + 120.0
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        stdout, stderr = child.communicate(timeout=total)
    except subprocess.TimeoutExpired:
        os.killpg(child.pid, signal.SIGKILL)
        stdout, stderr = child.communicate()
        stderr += f"\nbench.py: no result within {total:.0f} s; the rank processes were ended\n"
    sys.stderr.write(stderr)
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    for ln in stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    if lines:
        print(lines[-1], flush=True)
    return child.returncode if child.returncode is not None else 1


def pmc_child(args) -> int:
    """The process rocprofv3 wraps for `roofline.traffic`: the headline operator and its fp64-record twin, a short CG
    solve each (SpMV launches with the fused-dot epilogue, like the timed region's)."""
    import numpy as np

    from stormruler_amd import api, mesh

    g = mesh.structured_box(args.n)
    ctx = api.Context(0)
    for kv in args.opt:
        k_, v_ = kv.split("=")
        ctx.set_option(k_, int(v_))
    b = api.DeviceVector(ctx, g.n_cells)
    api.fill_with(b, 1.0)
    for dict_level in (None, 0):
        if dict_level is not None:
            ctx.set_option("spmv_dict", dict_level)
        mat = api.StencilMatrix.from_face_graph(ctx, g)
        s = api.CgSolver()
        s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = 12, 0.0, 0.0
        s.solve(api.DeviceVector(ctx, g.n_cells), b, api.HipStencilOperator(mat, -1.0, 0.0))
        ctx.sync()
        # ... and the stand-alone apply of the `spmv` block (no fused dot: a kernel instance of its own), rotating vectors
        xs = [api.DeviceVector.from_numpy(ctx, np.sin(0.37 * np.arange(g.n_cells, dtype=np.float64))) for _ in range(3)]
        ys = [api.DeviceVector(ctx, g.n_cells) for _ in range(3)]
        for i in range(9):
            mat.apply(-1.0, 0.0, xs[i % 3], ys[i % 3])
        ctx.sync()
        del xs, ys
        mat.close()
    if args.tet_prefix:  # the tetrahedral mesh of roofline_unstructured3d: a CG solve (fused dot) and stand-alone applies
        hm, mat, _ = tet_operator(api, ctx, args.tet_prefix)
        nt = mat.stats()["n_rows"]
        # (the kernel instances are the fp64-record ones of the box above: told apart by their grid size)
        bt = api.DeviceVector(ctx, nt)
        api.fill_with(bt, 1.0)
        s = api.CgSolver()
        s.num_iterations, s.absolute_error_tolerance, s.relative_error_tolerance = 12, 0.0, 0.0
        s.solve(api.DeviceVector(ctx, nt), bt, api.HipStencilOperator(mat, -1.0, 0.0))
        xs = [api.DeviceVector.from_numpy(ctx, np.sin(0.37 * np.arange(nt, dtype=np.float64))) for _ in range(3)]
        ys = [api.DeviceVector(ctx, nt) for _ in range(3)]
        for i in range(9):
            mat.apply(-1.0, 0.0, xs[i % 3], ys[i % 3])
        ctx.sync()
        mat.close()
    ctx.close()
    return 0


def measure_traffic(args, tet_prefix=None):
    """HBM bytes per SpMV launch by PMC, as /opt/skills/guides/MI355X_MICROARCH.md prescribes: FETCH_SIZE and WRITE_SIZE
    in SEPARATE rocprofv3 passes (--pmc with --kernel-trace only), FETCH_SIZE doubled (on gfx950 it reports exactly half the
    bytes of a wide coalesced streaming read: 128-byte requests tallied at 64 B), both in KiB.  Returns (headline kernel, fp64-record kernel,
    method note); (None, None, reason) when the profiler is unavailable -- never raises."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile

    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, None, "rocprofv3 not found", {}
    # Never from inside a profiled process: the inner launcher would inherit the outer profiler's preloaded tool
    # library, which initialises the GPU in the launcher before it execs the target -- the forbidden exec hop.
    def profiler_var(k, v):
        return (k.startswith(("ROCP_", "ROCPROF", "ROCPROFILER_", "ROCTRACER_", "HSA_TOOLS_")) or
                (k == "LD_PRELOAD" and any(t in v for t in ("rocprof", "roctracer", "rocprofiler"))))

    if any(profiler_var(k, v) for k, v in os.environ.items()):
        return None, None, "running under a profiler: no nested rocprofv3 (traffic not measured in this run)", {}
    sums = {}
    # the fp64-record kernel runs on the box AND on the tetrahedral mesh: told apart by the grid (threads) of the launch
    box_grid = ((args.n ** 3 + 255) // 256) * 256
    tmp = tempfile.mkdtemp(prefix="storm_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--",
                   sys.executable, os.path.abspath(__file__), "--pmc-child", "--edge", str(args.n),
                   *[f"--opt={kv}" for kv in args.opt], *(["--tet-prefix", tet_prefix] if tet_prefix else [])]
            env = {k: v for k, v in os.environ.items()
                   if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK") and not profiler_var(k, v)}
            env["TMPDIR"] = "/tmp"
            # (its own session: on a timeout the whole group goes -- the profiled python is a grandchild that would
            #  otherwise keep solving on the GPU under the measurements that follow)
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True,
                                    start_new_session=True)
            try:
                _, err = proc.communicate(timeout=90 + (90 if tet_prefix else 0))
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(proc.pid, signal.SIGKILL)
                except ProcessLookupError:
                    pass
                proc.communicate()
                return None, None, f"rocprofv3 --pmc {counter}: no result within its time limit (the process group was ended)", {}
            if proc.returncode != 0:
                return None, None, f"rocprofv3 --pmc {counter} exited with {proc.returncode}: {(err or '')[-300:]}", {}
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                with open(f, newline="") as fh:
                    for row in csv.DictReader(fh):
                        if row["Counter_Name"] != counter:
                            continue
                        k = row["Kernel_Name"]
                        if "spmv_sell_kernel<true" in k and row.get("Grid_Size") and box_grid is not None and int(row["Grid_Size"]) != box_grid:
                            a = sums.setdefault((counter, "tet_dot" if "spmv_sell_kernel<true, true" in k else "tet_plain"), [0, 0.0])
                            a[0] += 1
                            a[1] += float(row["Counter_Value"])
                            continue
                        tile = "spmv_canon_tile_kernel<true" in k
                        kind = ("step" if (tile and ", true>(" in k) or "cg_step_march_kernel" in k else  # the fused CG step (the dominant kernel of the headline run)
                                "fmt" if (tile or "spmv_canon_kernel<true" in k or "spmv_pair_kernel<true" in k or "spmv_dict_kernel<true" in k)
                                else "sell" if "spmv_sell_kernel<true, true" in k
                                # the stand-alone applies (no fused dot) of the `spmv` block
                                else "plain_lattice" if ("spmv_canon_tile_kernel<false" in k or "spmv_canon_kernel<false" in k or
                                                         "spmv_pair_kernel<false" in k or "spmv_dict_kernel<false" in k)
                                else "plain_sell" if "spmv_sell_kernel<true, false" in k else None)
                        if kind:
                            a = sums.setdefault((counter, kind), [0, 0.0])
                            a[0] += 1
                            a[1] += float(row["Counter_Value"])

        def per_launch(kind):
            f, w = sums.get(("FETCH_SIZE", kind)), sums.get(("WRITE_SIZE", kind))
            if not f or not w:
                return None
            return (2.0 * f[1] / f[0] + w[1] / w[0]) * 1024.0

        step, fmt, sell = per_launch("step"), per_launch("fmt"), per_launch("sell")
        head = "step" if step is not None else "fmt"
        note = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over a child process "
                f"(12-iteration CG, {sums.get(('FETCH_SIZE', head), [0])[0]} launches of the "
                f"{'fused CG step kernel' if head == 'step' else 'SpMV kernel'} averaged); FETCH_SIZE doubled (gfx950), KiB"
                + (f"; the plain SpMV launch of the same operator: {fmt:.0f} B" if step is not None and fmt is not None else ""))
        plain = {"lattice": per_launch("plain_lattice"), "general": per_launch("plain_sell"),
                 "tet": per_launch("tet_plain"), "tet_dot": per_launch("tet_dot"),
                 "launches": {"lattice": sums.get(("FETCH_SIZE", "plain_lattice"), [0])[0],
                              "general": sums.get(("FETCH_SIZE", "plain_sell"), [0])[0]}}
        return (step if step is not None else fmt if fmt is not None else sell), sell, note, plain
    except Exception as e:  # the measurement must never cost the headline line
        return None, None, f"traffic measurement failed: {e!r}", {}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def blas1_rates(api, ctx, N, reps=20):
    """Algorithmic GB/s of each BLAS-1 statement of the solver bodies at N elements (SURVEY.md 8d byte counts),
    HIP events around `reps` back-to-back calls on the library's stream.  dot / norm2 / multi_dot return their
    result to the host, so their figure includes the host round trip between two calls (the kernel's last block
    writes the sums to pinned memory, the host polls them and launches again); the "in flight" entries are the same
    kernels without that gap."""
    from stormruler_amd._lib import check, lib

    v = [api.DeviceVector(ctx, N) for _ in range(10)]
    for i, w in enumerate(v):
        api.fill_with(w, 1.0 + 0.001 * i)
    a, b, c = v[0], v[1], v[2]

    def t(fn, bytes_per_elt):
        for _ in range(3):
            fn()
        ctx.timer_start()
        for _ in range(reps):
            fn()
        ms = ctx.timer_stop() / reps
        gbs = bytes_per_elt * N / (ms * 1e-3) / 1e9
        return {"ms": ms, "bytes_per_element": bytes_per_elt, "GBs": gbs, "frac_of_peak": gbs / HBM_PEAK_GBS}

    out = {"n": N}
    out["copy (a <<= b)"] = t(lambda: a.__ilshift__(b), 16)
    out["fill"] = t(lambda: api.fill_with(a, 1.0), 8)
    out["scale (a *= s)"] = t(lambda: a.__imul__(1.0000001), 16)
    out["axpy (a += s b)"] = t(lambda: a.__iadd__(1e-9 * b), 24)
    out["xpay (a <<= b + s a)"] = t(lambda: a.__ilshift__(b + 0.5 * a), 24)
    out["sub (a <<= b - c)"] = t(lambda: a.__ilshift__(b - c), 24)
    out["bicgstab p (a <<= b + s (a - w c))"] = t(lambda: check(lib.storm_hip_bicgstab_p(a._h, b._h, 0.5, 0.25, c._h)), 32)
    # the element map of the playground's time loop (`f <<= map(dF_dc, c)`, Playground.cpp:148): a traced 13-operation program
    dF = api.map(lambda c_: 2.0 * c_ * (c_ - 1.0) * (2.0 * c_ - 1.0), b)
    out["map dF_dc (a <<= map(f, b))"] = t(lambda: a.__ilshift__(dF), 16)
    out["dot"] = t(lambda: api.dot_product(a, b), 16)
    out["norm2"] = t(lambda: api.norm_2(a), 8)
    out["multi_dot k=8"] = t(lambda: api.multi_dot(a, v[1:9]), 8 * 9)
    # the same kernels with the host round trip between two synchronous calls (~9 us: result to the host, the next
    # launch) taken out: storm_hip_multi_dot_begin / _end, four requests in flight -- the kernels' own rate
    pending = []

    def in_flight(bs):
        pending.append(api.PendingDots(a, bs))
        if len(pending) == 4:
            pending.pop(0).result()

    out["dot, 4 in flight (begin / end)"] = t(lambda: in_flight([b]), 16)
    out["norm2, 4 in flight (begin / end)"] = t(lambda: in_flight([a]), 8)
    while pending:
        pending.pop(0).result()
    out["multi_axpy k=8"] = t(lambda: api.multi_axpy(a, [1e-9] * 8, v[1:9]), 8 * 10)
    return out


def cpu_baseline(args, n, g, perm, run, ctx, samples=None):
    """`cpu_samples` (the CPU-only part, possibly run on a thread beside the GPU-side blocks) + the GPU-side checks."""
    cpu, r = samples if samples is not None else cpu_samples(args, n, g, perm)
    return cpu_checks(args, cpu, r, run, ctx)


def cpu_samples(args, n, g, perm):
    """The oracle (port of the reference path) on the GPU box's host cores: 1 thread, bounded sample.  `value` is the
    strict build (the parity checker: gcc -O2 -ffp-contract=off, built in the container); beside it, when this machine has a
    C compiler, SURVEY 8d's flags built HERE: -O3 -march=native and the same with -ffast-math (the reference's Release is
    -Ofast -march=native, CMakeLists.txt:194-195).  Labelled extras -- never the checker."""
    import numpy as np

    from oracle import oracle
    from stormruler_amd import api, mesh

    g_cpu = g if perm is None else mesh.structured_box(n)
    ones = np.ones(g_cpu.n_cells)

    extra_iters = max(args.cpu_iters // 2, 1)  # the labelled extras: half the sample (the bench's wall time)

    def sample(variant, iters=None):
        iters = iters or extra_iters
        o_ = oracle.StencilOperator(g_cpu, -1.0, 0.0, variant=variant)
        t_ = time.perf_counter()
        r_ = oracle.solve("cg", o_, ones, num_iterations=iters, abs_tol=0.0, rel_tol=0.0, variant=variant)
        return r_, time.perf_counter() - t_

    r, tc = sample("strict", args.cpu_iters)
    # the init apply counts as work: iterations + 1 applies were done
    cpu = {"value": args.cpu_iters / tc, "unit": "iter/s", "cores": 1, "kind": "port",
           "sample": f"{args.cpu_iters} CG iterations (+ init residual) of the same {n}^3 problem; oracle/liboracle.so, gcc -O2 "
                     f"-ffp-contract=off (the parity checker's build); 1 thread of {os.cpu_count()} cpus",
           "sample_long": "Flags of `value`: gcc -O2 -ffp-contract=off -- the reference's statement order with strict IEEE rounding, what "
                          "the GPU path is compared with; built in a container that does not know this box's CPU.  `value_fma_build`: "
                          "the same source at -O3 -mavx2 -mfma -ffp-contract=fast, prebuilt.  `value_native_O3` / "
                          "`value_native_O3_fast_math`: SURVEY 8d's flags (gcc -O3 -march=native, and with -ffast-math: the reference's "
                          "Release is -Ofast -march=native, CMakeLists.txt:194-195), compiled on THIS machine when it has a compiler; "
                          "fast-math reassociates the sums -- a different rounding, so never the checker",
           "seconds": tc, "extras_sample_iterations": extra_iters}
    try:
        cpu["value_fma_build"] = extra_iters / sample("fma")[1]
    except Exception:
        pass
    for variant, key in (("native", "value_native_O3"), ("native_fast", "value_native_O3_fast_math")):
        try:
            rn, tn = sample(variant)
            cpu[key] = extra_iters / tn
            ref_half = float(r.history[extra_iters])  # (history[0] is the initial residual: entry k = after k iterations)
            cpu[key + "_residual_rel_diff_vs_strict"] = abs(rn.absolute_error - ref_half) / ref_half
            cpu["native_flags"] = "built on this machine: " + oracle.native_flags("native") + " [-ffast-math]"
        except Exception as e:
            cpu[key + "_error"] = repr(e)[:200]
    # "What the host CPU could do" (optional; --cpu-parallel): OpenMP CG on the same box -- NOT the
    # reference's algorithm order (gather SpMV on assembled rows, parallel reductions; oracle/storm_oracle_omp.c), reported
    # beside the faithful single-threaded port, never instead of it.  Min / median of 3 runs: one run is noise-dominated.
    if args.cpu_parallel:
        try:
            ncpu = os.cpu_count() or 1
            rates, used, res = [], None, None
            th = min(ncpu, 64)
            for _ in range(3):
                res, sec, used = oracle.omp_cg_box(n, args.cpu_iters, th)
                rates.append(args.cpu_iters / sec)
            cpu["parallel"] = {"value": float(np.median(rates)), "value_min": min(rates), "value_max": max(rates), "unit": "iter/s",
                               "cores": used, "residual_rel_diff_vs_port": abs(res - r.absolute_error) / r.absolute_error,
                               "kind": "OpenMP port, gather SpMV + parallel reductions (not the reference's single-threaded loop order)",
                               "sample": f"{args.cpu_iters} CG iterations of the same {n}^3 problem, {th} threads, median of 3 runs"}
        except Exception as e:
            cpu["parallel"] = {"error": repr(e)}
    return cpu, r


def cpu_checks(args, cpu, r, run, ctx):
    """The GPU side of `cpu_baseline`: the same sample on the device, and BASELINE config 1 on both."""
    import numpy as np

    from oracle import oracle
    from stormruler_amd import api, mesh

    # parity spot check at bench size: same iteration count of CG from the same start gives the
    # same residual (GPU sums in a different order: tolerance, not bits)
    sg, _ = run(args.cpu_iters)
    cpu["gpu_vs_cpu_residual_rel_diff"] = abs(sg.absolute_error - r.absolute_error) / r.absolute_error
    # BASELINE config 1 (the reference's CPU-runnable case): 64^3, full solve to the default
    # tolerances, CPU oracle vs this library -- iteration counts and solutions must agree
    g64 = mesh.structured_box(64)
    t64 = time.perf_counter()
    r64 = oracle.solve("cg", oracle.StencilOperator(g64, -1.0, 0.0), np.ones(g64.n_cells))
    t64 = time.perf_counter() - t64
    m64 = api.StencilMatrix.from_face_graph(ctx, g64)
    b64, x64 = api.DeviceVector(ctx, g64.n_cells), api.DeviceVector(ctx, g64.n_cells)
    api.fill_with(b64, 1.0)
    op64 = api.HipStencilOperator(m64, -1.0, 0.0)
    api.CgSolver().solve(api.DeviceVector(ctx, g64.n_cells), b64, op64)  # untimed: first use loads the kernel's code object
    s64 = api.CgSolver()
    ctx.sync()
    tg = time.perf_counter()
    s64.solve(x64, b64, op64)
    ctx.sync()
    tg = time.perf_counter() - tg
    xg = x64.to_numpy()
    cpu["config1_64cubed"] = {
        "cpu_iterations": r64.iterations, "gpu_iterations": s64.iteration, "cpu_seconds": t64, "gpu_seconds": tg,
        "gpu_path": "latency path: one cooperative persistent kernel per solve (csrc/latency.hip)",
        "solution_rel_diff": float(np.linalg.norm(xg - r64.x) / np.linalg.norm(r64.x))}
    m64.close()

    return cpu


if __name__ == "__main__":
    sys.exit(main())
