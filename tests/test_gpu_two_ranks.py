"""The multi-rank device path with several ranks on ONE GPU (multi-GPU-vs-1-GPU parity).

`gpurun` boxes have one MI355X and RCCL refuses two ranks on the same device, so the ranks here use the library's
host-staged transport (`storm_hip_ctx_comm_init_host`: halo planes and reduction scalars travel through gloo) --
the partition, the halo plans, the interior/boundary split of the SpMV, the placement of every all-reduce in the
three device-resident solver loops and the rank-consistent convergence decision are exactly the ones of the
8-GPU run; only the bytes take another road.  The RCCL calls themselves are covered by tests/test_gpu_comm.py."""
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


@pytest.mark.parametrize("world,dims,transport", [(3, (20, 8, 5), "ipc"), (4, (12, 12, 4), "ipc"), (2, (64, 16, 12), "host")])
def test_partitioned_device_path_matches_the_global_oracle(world, dims, transport, tmp_path):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "multi_rank_worker.py"), *map(str, dims)]
    env = dict(os.environ, OMP_NUM_THREADS="1", STORM_REPORT_DIR=str(tmp_path), STORM_TRANSPORT=transport,
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    reports = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    assert sorted(r["rank"] for r in reports) == list(range(world))
    for r in reports:  # interior ranks talk to two neighbours, the end ranks to one
        assert len(r["nbrs"]) == (1 if r["rank"] in (0, world - 1) else 2)


def test_cavity_on_four_ranks(tmp_path):
    """BASELINE config 5's shape -- lid-driven cavity, pressure-Poisson CG each step, 4 ranks -- on one GPU."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "multi_rank_cavity_worker.py"), "16", "4"]
    env = dict(os.environ, OMP_NUM_THREADS="1", STORM_REPORT_DIR=str(tmp_path))
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    reports = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(4)]
    assert all(len(r["steps"]) == 4 for r in reports)
    # every rank saw the same CG iteration counts
    assert all(r["steps"][k][0] == reports[0]["steps"][k][0] for r in reports for k in range(4))


def test_peer_window_waits_are_bounded(tmp_path):
    """A rank that never joins a reduction must not hang the others: the waiting kernel gives up after 5 s and the next
    library call returns STORM_HIP_E_COMM (-4)."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "multi_rank_timeout_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1", STORM_REPORT_DIR=str(tmp_path), HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    r0 = json.load(open(tmp_path / "rank0.json"))
    assert r0["outcome"] == "error" and r0["status"] == -4 and "timed out" in r0["what"], r0
    assert 4.0 <= r0["seconds"] <= 9.0, r0


def _run_bench(extra, timeout=420, launcher_ranks=0):
    # launcher_ranks = 0: without a launcher in front -- bench.py starts its rank supervisors itself, each of which starts
    # the measuring rank as a child; launcher_ranks = N: exactly as the driver calls it for N > 1,
    # `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`
    # (every launched process is then the supervisor of its rank)
    front = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={launcher_ranks}", "--master-addr",
             "127.0.0.1", "--master-port", str(_free_port())] if launcher_ranks else [sys.executable]
    import tempfile

    with tempfile.TemporaryDirectory() as tmp:
        detail_path = os.path.join(tmp, "bench_detail.json")
        cmd = [*front, os.path.join(ROOT, "bench.py"), "--shared-device", "--edge", "48", "--steps", "20", "--warmup", "3",
               "--spinup-seconds", "0.2", "--min-seconds", "0.05", "--detail-path", detail_path, *extra]
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
        p = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout,
                           env=dict(env, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0"), cwd=ROOT)
        assert p.returncode == 0, p.stdout[-2000:] + "\n" + p.stderr[-3000:]
        detail = json.load(open(detail_path))
    detail["_line"] = _the_one_compact_line(p.stdout)
    return detail, p.stderr


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline")


def _the_one_compact_line(stdout):
    """stdout holds ONE JSON line, short enough for the driver's 8 KB tail (BENCH_r05.json: a 21 KB line was not parsed),
    strict JSON, with the contract's fields; everything else is in the detail file."""
    lines = [ln for ln in stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and stdout.rstrip().endswith(lines[0])
    assert len(lines[0]) <= 8000, len(lines[0])
    line = json.loads(lines[0], parse_constant=lambda c: pytest.fail(f"non-strict JSON constant {c}"))
    for key in CONTRACT:
        assert key in line, key
    assert "dropped" not in line
    r = line["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and 0.0 < r["frac"] <= 1.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-4
    return line


@pytest.mark.parametrize("world,transport", [(3, "ipc")])  # (2 ranks, host AND ipc: test_bench_two_ranks_on_the_production_kernels_measures_both_transports)
def test_bench_script_multi_rank_path(world, transport):
    """bench.py's own N > 1 code path (supervisors, pre-flight, partition, connect, the rank-uniform spin-up, barriers,
    max-over-ranks timing, rank-0 JSON) with the ranks sharing device 0 (`--shared-device`): it must terminate -- a
    collective that only some ranks reach hangs the 8-GPU run -- and print one JSON line on rank 0."""
    out, _ = _run_bench(["--gpus", str(world), "--transport", transport], launcher_ranks=world)  # (the driver's own command line)
    assert out["n_gpus"] == world and out["steps"] == 20 and out["value"] > 0 and out["scaling"] == "weak"
    assert out["transport"] == transport and out["transport_fallback"] == []
    assert out["preflight"]["ok"] and out["preflight"]["allreduce_sum"] == out["preflight"]["expected_sum"]
    assert out["postflight"]["ok"] and out["postflight"]["halo_rows_wrong_on_this_rank"] == 0
    assert out["postflight"]["fused_vs_unfused_residual_rel_diff"] <= 1e-9
    assert out["roofline"]["launches_timed"] >= 20 + 1  # one launch per apply, or an interior + a boundary launch
    assert 0.0 < out["roofline"]["frac"] <= 1.0 and out["timing"]["repeats"] >= 1
    line = out["_line"]
    assert line["n_gpus"] == world and line["transport"] == transport and line["preflight_ok"] and line["postflight_ok"]
    assert abs(line["value"] - out["value"]) <= 1e-5 * out["value"]
    # (ranks sharing one device on the peer windows: no RCCL communicator -- RCCL's own view says so)
    assert line["rccl"]["nccl_comm_count"] == 0 and len(line["rccl"]["ranks"]) == world
    assert line["transports_measured"][transport]["comm_breakdown"]["allreduces_per_iteration"] > 1.9
    # where the communication time went: the peer-window transport's kernels time their own waits
    cb = out["comm_breakdown"]
    assert cb["transport"] == transport
    if transport == "ipc":
        assert 1.9 <= cb["allreduces_per_iteration"] <= 2.3  # CG: <p, Ap> and <r, r> (+ the solves' init residuals)
        assert 0.0 < cb["allreduce_wait_us_per_iteration_worst_rank"] < 1e4
        assert cb["send_ack_wait_us_per_iteration_worst_rank"] >= 0.0


@pytest.mark.parametrize("how", ["exit:1", "hang:0", "wrong"])  # (a rank dies / hangs / the pre-flight finds wrong values; "post:1", a
# wrong value found by the post-flight, takes the same road after the timed region: run by hand, `--inject-fail ipc=post:1`)
def test_bench_falls_back_to_the_next_transport_with_fresh_ranks(how):
    """A transport that fails -- a rank dies, a rank hangs (budget), or the pre-flight finds wrong halo values -- costs
    its budget, not the run: all rank processes of the attempt are ended and a FRESH set starts on the next
    transport of the chain; the line says which transport produced the number and why the earlier one did not."""
    # (a hanging rank costs the whole budget of its attempt: a short one for that case)
    out, err = _run_bench(["--gpus", "2", "--transport", "ipc,host", "--inject-fail", f"ipc={how}",
                           "--attempt-seconds", "5,240" if how.startswith("hang") else "45,240"])
    assert out["transport"] == "host" and out["n_gpus"] == 2 and out["value"] > 0
    fb = out["transport_fallback"]
    assert len(fb) == 1 and fb[0]["transport"] == "ipc"
    assert ("budget" in fb[0]["reason"]) == how.startswith("hang")
    assert "starting fresh ranks on host" in err


@pytest.mark.parametrize("world", [3])  # (3 parts of the RCB are not a chain of slabs; 4 ranks: test_cavity_on_four_ranks)
def test_unstructured_partition_reproduces_the_recorded_reference_run(world, tmp_path):
    """General meshes: RCB parts of the reference's Triangle mesh, general halo plans."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "multi_rank_unstructured_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1", STORM_REPORT_DIR=str(tmp_path))
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert p.returncode == 0, p.stdout[-3000:] + "\n" + p.stderr[-3000:]
    reports = [json.load(open(tmp_path / f"rank{r}.json")) for r in range(world)]
    assert sum(r["n_local"] for r in reports) == 6252
    assert len({r["iterations"] for r in reports}) == 1
    assert max(len(r["nbrs"]) for r in reports) >= 2  # not a chain of slabs


def test_bench_line_contract_at_one_gpu(tmp_path):
    """`python bench.py` (N = 1) on a small edge: ONE JSON line carrying the contract's fields -- `roofline` with a
    physical fraction <= 1 that follows from its own bytes and time, `roofline_general`, `cpu_baseline`, `timing`."""
    detail_path = os.path.join(str(tmp_path), "bench_detail.json")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--edge", "96", "--steps", "20", "--warmup", "3",
           "--spinup-seconds", "0.2", "--min-seconds", "0.05", "--cpu-iters", "4", "--tet-edge", "12", "--self-exchange", "cg",
           "--detail-path", detail_path]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=dict(env, OMP_NUM_THREADS="1"), cwd=ROOT)
    assert p.returncode == 0, p.stdout[-2000:] + "\n" + p.stderr[-3000:]
    line = _the_one_compact_line(p.stdout)
    out = json.load(open(detail_path))  # the full record
    # ---- the compact line: the contract + roofline (with the 8d-comparable SpMV figure inside) + cpu_baseline ----
    assert line["n_gpus"] == 1 and line["steps"] == 20 and line["dtype"] == "f64" and line["vs_baseline"] is None
    assert "workload" in line["config"] and "model" not in line["config"] and line["detail"]
    assert abs(line["value"] - out["value"]) <= 1e-5 * out["value"] and abs(line["value_general"] - out["value_general"]) <= 1e-5 * out["value_general"]
    lr = line["roofline"]
    assert abs(lr["spmv_general_frac_8d_rotating"] - out["spmv"]["general"]["rotating_3_pairs"]["frac_8d"]) <= 1e-5
    assert abs(lr["spmv_general_ms"] - out["spmv"]["general"]["rotating_3_pairs"]["median_ms"]) <= 1e-5 * lr["spmv_general_ms"]
    assert lr["spmv_general_bytes_8d"] == out["spmv"]["general"]["algorithmic_bytes_8d"] and lr["kernel"]
    lc = line["cpu_baseline"]
    assert lc["kind"] == "port" and lc["cores"] == 1 and lc["value"] > 0 and len(lc["sample"]) <= 200 and lc["unit"] == "iter/s"
    for k in ("config1_cg64", "config3_bicgstab256", "config4_gmres30_convdiff128", "config5_cavity128"):
        assert k in line["configs"] and "error" not in line["configs"][k], line["configs"]
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert key in out, key
    assert out["n_gpus"] == 1 and out["steps"] == 20 and out["dtype"] == "f64" and out["vs_baseline"] is None
    assert "workload" in out["config"] and "model" not in out["config"]
    for name in ("roofline", "roofline_general"):
        r = out[name]
        assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
        assert 0.0 < r["frac"] <= 1.0
        assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-12
        assert abs(r["achieved"] - r["bytes_per_launch"] / (r["avg_launch_ms"] * 1e-3) / 1e9) <= 1e-6 * r["achieved"]
    assert out["roofline_general"]["bytes_per_launch"] >= out["roofline_general"]["algorithmic_bytes_8d"]
    assert "plain_spmv_frac" not in out["roofline"] and out["roofline"]["frac_8d"] > 0
    # the SpMV alone, SURVEY 8d: >= 50 stand-alone applies, median and mean, three byte counts, two access patterns
    for fmt in ("lattice", "general"):
        b = out["spmv"][fmt]
        assert b["algorithmic_bytes_8d"] == 24 * b["rows"] + 12 * b["nnz_offdiag"] and b["streamed_bytes"] > 0
        for mode in ("back_to_back", "rotating_3_pairs"):
            m = b[mode]
            assert m["launches"] >= 50 and 0 < m["min_ms"] <= m["median_ms"] <= m["max_ms"] and m["mean_ms"] > 0
            assert abs(m["frac_8d"] - b["algorithmic_bytes_8d"] / (m["median_ms"] * 1e-3) / 1e9 / 8000.0) <= 1e-9
            assert abs(m["frac_streamed"] - b["streamed_bytes"] / (m["median_ms"] * 1e-3) / 1e9 / 8000.0) <= 1e-9
    assert out["spmv"]["general"]["streamed_bytes"] >= out["spmv"]["general"]["algorithmic_bytes_8d"]
    # a genuinely unstructured 3-D mesh: tetrahedra through the TetGen files and the library's reader
    u = out["roofline_unstructured3d"]
    assert u["rows"] == 6 * 12 ** 3 and u["max_row_len"] == 4 and u["tail_nnz"] == 0 and u["record_format"].startswith("fp64")
    assert u["nnz_offdiag"] == 2 * u["interior_faces"] and 0 <= u["ell_padding_ratio"] < 0.2
    assert 0.0 < u["frac"] <= 1.0 and u["cg_iter_per_s"] > 0 and u["spmv"]["rotating_3_pairs"]["launches"] >= 50
    # a user's statement-by-statement CG against the library's device loop (child processes: the C++ driver)
    h = out["host_loop_cg256"]
    assert "error" not in h, h
    assert 0 < h["device_loop_us_per_iteration"] <= h["host_loop_lazy_statements_us_per_iteration"] * 1.5
    assert h["host_loop_lazy_statements_us_per_iteration"] < h["host_loop_eager_statements_us_per_iteration"]
    # the multi-rank code path at one rank: a size-1 RCCL communicator, the planes exchanged with the rank itself
    mr = out["multi_rank_path_at_one_rank"]
    assert "error" not in mr, mr
    assert mr["cg"]["us_per_iteration_over_rccl"] > 0 and abs(mr["cg"]["us_per_iteration_plain"] - out["ms_per_step"] * 1e3) <= 1e-9
    br = mr["cg"]["comm_breakdown"]
    assert br["transport"] == "rccl" and 0.9 <= br["halo_exchanges_per_iteration"] <= 1.1 and 1.9 <= br["allreduces_per_iteration"] <= 2.1
    cpu = out["cpu_baseline"]
    assert cpu["kind"] == "port" and cpu["cores"] == 1 and cpu["value"] > 0 and cpu["gpu_vs_cpu_residual_rel_diff"] <= 1e-9
    # SURVEY 8d's flags, built on THIS machine (it has gcc: the tests' C hosts are built with it), beside the strict build
    assert cpu["value_native_O3"] > 0 and cpu["value_native_O3_fast_math"] > 0 and lc["value_native_O3"] > 0
    assert cpu["value_native_O3_residual_rel_diff_vs_strict"] <= 1e-9 and cpu["value_native_O3_fast_math_residual_rel_diff_vs_strict"] <= 1e-6
    assert out["timing"]["repeats"] >= 1 and out["value"] > 10 * cpu["value"]


def test_bench_two_ranks_on_the_production_kernels_measures_both_transports():
    """Two ranks (sharing the one GPU) at 128^3 per rank: large enough for the lattice kernels -- tiled interior launch with
    the sending blocks, the marching fused CG step, the boundary launch that reads the window -- so the multi-process
    windows carry exactly what an N-GPU run does; the post-flight check (halo test at full size + fused against unfused
    residual) must pass and no fallback may have happened.  And N > 1 measures every data-path transport of the chain by
    fresh rank processes (on N devices: RCCL and the peer windows; here, ranks sharing one device: host-staged and the peer
    windows): `value` is the better one, `transports_measured` holds both with their breakdowns."""
    out, _ = _run_bench(["--gpus", "2", "--transport", "host,ipc", "--edge", "128", "--steps", "30"], timeout=600)
    tm = out["transports_measured"]
    assert set(tm) == {"host", "ipc"} and out["transport_fallback"] == []
    best = max(tm, key=lambda t: tm[t]["value"])
    assert out["transport"] == best == "ipc" and out["value"] == tm[best]["value"] and out["ms_per_step"] == tm[best]["ms_per_step"]
    for t in tm.values():
        assert t["value"] > 0 and t["postflight"]["ok"] and "comm_breakdown" in t
    assert out["postflight"]["ok"] and out["postflight"]["fused_vs_unfused_residual_rel_diff"] <= 1e-9
    assert out["op_stats"]["tiled_planes"] == 2 and out["op_stats"]["paired_rows"] == 2
    assert out["roofline"]["launches_timed"] >= 2 * 31  # per apply: the interior / marching launch + the boundary launch
    line = out["_line"]
    assert set(line["transports_measured"]) == {"host", "ipc"} and line["transport"] == "ipc" and line["postflight_ok"]
    assert line["transports_measured"]["ipc"]["value"] == pytest.approx(tm["ipc"]["value"], rel=1e-5)

def test_bench_contact_mode_over_rccl_at_one_rank_and_on_two_ranks(tmp_path):
    """`--contact`: what a short lease on N devices runs -- edge 64, pre-flight, one CG and one
    BiCGStab over the transport with the device-timestamp breakdown, the compact line.  Here (one device): over RCCL on a
    size-1 communicator (`--force-comm`: RCCL's own view must say 1 rank on this device), and on two ranks sharing the
    device over the peer windows.  Both within the budget of a short lease."""
    import time

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env = dict(env, OMP_NUM_THREADS="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    detail = os.path.join(str(tmp_path), "contact.json")
    t0 = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--contact", "--force-comm", "--steps", "10", "--detail-path", detail],
                       capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    wall = time.time() - t0
    assert p.returncode == 0, p.stdout[-2000:] + "\n" + p.stderr[-3000:]
    line = _the_one_compact_line(p.stdout)
    assert line["n_gpus"] == 1 and line["steps"] == 10 and line["config"]["cells_per_gpu"] == 64 ** 3
    assert line["rccl"]["nccl_comm_count"] == 1 and line["rccl"]["distinct_devices"] == 1 and line["rccl"]["ranks"][0][:3] == [0, 0, 0]
    assert line["transport"] == "rccl" and line["comm_breakdown"]["transport"] == "rccl" and line["comm_breakdown"]["allreduces_per_iteration"] > 1.9
    cb = line["contact_bicgstab"]
    assert "error" not in cb and cb["iterations"] == 10 and cb["us_per_iteration"] > 0
    assert cb["comm_breakdown"]["allreduces_per_iteration"] >= 2.9 and cb["comm_breakdown"]["allreduce_us_each"] > 0
    assert wall < 90, wall  # (python + torch start-up included: the GPU work is a fraction of a second)
    # two ranks (sharing the device: peer windows), through the supervisors
    out, _ = _run_bench(["--gpus", "2", "--contact", "--steps", "10"], timeout=300)
    line = out["_line"]
    assert line["n_gpus"] == 2 and line["transport"] == "ipc" and line["preflight_ok"] and line["postflight_ok"]
    assert line["contact_bicgstab"]["iterations"] == 10 and len(line["rccl"]["ranks"]) == 2
