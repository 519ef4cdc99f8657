"""bench.py's stdout contract: ONE compact JSON line the driver can parse (<= 8000 characters -- the driver keeps ~8 KB of
stdout tail; BENCH_r05.json had `parsed: null` because the line had grown to 21 KB), carrying the contract's fields,
`roofline` (with the SURVEY-8d-comparable fp64-record SpMV figure inside it) and `cpu_baseline`; everything else goes to
bench_detail.json.  CPU only: the line is assembled from worst-case records, nothing is measured."""
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import bench  # noqa: E402

CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
LONG = "x" * 4000  # a note / an error text / a workload description that got out of hand


def _spmv_block(kernel):
    mode = {"launches": 60, "median_ms": 0.2533024996519089, "mean_ms": 0.2534092677136262, "min_ms": 0.24, "max_ms": 0.259,
            "GBs_8d_bytes": 6339.827464027547, "frac_8d": 0.7924784330034433, "GBs_streamed_bytes": 6346.036893473215,
            "frac_streamed": 0.7932546116841519, "GBs_pmc_traffic": 6495.456140626368, "frac_pmc_traffic": 0.811932017578296}
    return {"kernel": kernel, "record_format": LONG, "input": LONG, "rows": 16777216, "nnz_offdiag": 100270080,
            "algorithmic_bytes_8d": 1605894144, "streamed_bytes": 1607467008, "pmc_traffic_bytes": 1645315276.8,
            "back_to_back": dict(mode), "rotating_3_pairs": dict(mode), "all_launches_mean_ms": 0.2498}


def _breakdown():
    return {"transport": "rccl", "iterations_covered": 400, "halo_exchanges_per_iteration": 2.0025, "allreduces_per_iteration": 3.0025,
            "event_to_comm_stream_us_each_worst_rank": 3.1549937578027465, "pack_us_each_worst_rank": 6.309862671660424,
            "sendrecv_us_each_worst_rank": 80.47595505617977, "halo_unhidden_wait_us_each_worst_rank": 0.0022347066167290887,
            "halo_done_to_boundary_rows_us_each_worst_rank": 3.3189637952559305, "allreduce_us_each_worst_rank": 1.4772356369691924,
            "halo_unhidden_us_per_iteration_worst_rank": 6.6507000000000005, "allreduce_us_per_iteration_worst_rank": 4.4354000000000005,
            "note": LONG}


def worst_case_record(n_gpus):
    """Every block bench.py can put into a record, at its longest: all optional measurements present, long texts, and for
    N > 1 all four transports measured with breakdowns, fallbacks, and `n_gpus` ranks' worth of RCCL view."""
    full = {
        "metric": "CG iterations/sec, 256^3 Poisson per GPU (+ SpMV achieved HBM GB/s in `roofline`)", "value": 4583.061234567 * n_gpus,
        "unit": "iter/s", "n_gpus": n_gpus, "steps": 20, "warmup": 5, "ms_per_step": 0.21819484500156, "higher_is_better": True,
        "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": LONG, "cells_per_gpu": 16777216, "interior_faces_per_gpu": 50135040, "ordering": "natural",
                   "partition": LONG, "value_definition": LONG},
        "roofline": {"kernel": "cg_step_march_kernel (" + LONG + ")", "fused_cg_step": True, "bound": "hbm", "achieved": 6227.82286522427,
                     "peak": 8000.0, "unit": "GB/s", "frac": 0.7784778581530337, "frac_8d": 2.2203126858371403, "traffic": 961182984.7272727,
                     "traffic_method": LONG, "traffic_from_profile": {"file": LONG, "note": LONG}, "bytes_per_launch": 939524096,
                     "record_format": LONG, "avg_launch_ms": 0.15085915517061949, "launches_timed": 201, "measured_copy_GBs": 6280.229431947785,
                     "note": LONG},
        "spmv": {"lattice": _spmv_block("spmv_canon_tile_kernel"), "general": _spmv_block("spmv_sell_kernel")},
        "roofline_general": {"frac": 0.74966, "frac_8d": 0.74966, "kernel": LONG},
        "roofline_permuted_rcm": {"ordering": LONG, "frac": 0.7}, "roofline_unstructured": {"geometry": LONG, "frac": 0.7},
        "roofline_unstructured3d": {"mesh": LONG, "frac_8d": 0.7531864065801122, "traffic_over_8d_bytes": 1.1884137891314381,
                                    "spmv": _spmv_block("spmv_sell_kernel")},
        "config1_cg64": {"workload": LONG, "us_per_iteration": 8.386531234}, "config3_bicgstab256": {"error": LONG},
        "config4_gmres30_convdiff128": {"workload": LONG, "us_per_inner_iteration": 77.7488, "frac": 0.552956},
        "config5_cavity128": {"workload": LONG, "s_per_step": 0.00758501, "s_per_step_all": [0.0075] * 6},
        "extra_gmres30_poisson256": {"us_per_inner_iteration": 1096.58, "frac": 0.711938},
        "host_loop_cg256": {"workload": LONG, "host_loop_lazy_over_device_loop": 1.04963},
        "multi_rank_path_at_one_rank": {"workload": LONG, "cg": {"overhead_us_per_iteration": 12.7062, "comm_breakdown": _breakdown()},
                                        "bicgstab": {"overhead_us_per_iteration": 36.47, "comm_breakdown": _breakdown()}},
        "value_general": 2278.18123456, "general_mesh_path": {"record_format": LONG},
        "blas1": dict({"n": 16777216}, **{f"statement number {i} (a <<= b + s (a - w c))": {"ms": 0.0857, "bytes_per_element": 32, "GBs": 6263.59,
                                                                                             "frac_of_peak": 0.7829494246497337} for i in range(13)}),
        "cpu_baseline": {"value": 3.7548742116088047, "unit": "iter/s", "cores": 1, "kind": "port", "sample": LONG, "sample_long": LONG,
                         "seconds": 5.3, "gpu_vs_cpu_residual_rel_diff": 3.984497084032892e-12, "value_fma_build": 3.7568541667624866,
                         "value_native_O3": 4.1, "value_native_O3_fast_math": 4.4, "native_flags": LONG,
                         "parallel": {"value": 91.75771000336141, "value_min": 80.1, "cores": 32, "kind": LONG, "sample": LONG},
                         "config1_64cubed": {"cpu_iterations": 129, "gpu_iterations": 129, "cpu_seconds": 0.4867281750048278,
                                             "gpu_seconds": 0.0011672730033751577, "gpu_path": LONG, "solution_rel_diff": 1.0340048705055776e-12}},
        "timing": {"repeats": 6, "timed_seconds_total": 0.26, "ms_per_step_min": 0.218109, "ms_per_step_max": 0.218937, "ms_per_step_median": 0.218195},
        "cg": {"final_residual": 2132.3412345678, "note": LONG}, "op_stats": {k: 123456789 for k in "abcdefghijklmnop"},
        "device": "AMD Instinct MI355X", "setup_seconds": 3.2,
    }
    if n_gpus > 1:
        full["transport"] = "rccl"
        full["preflight"] = {"ok": True, "note": LONG}
        full["postflight"] = {"ok": True, "fused_vs_unfused_residual_rel_diff": 1e-13}
        full["rccl_view"] = [{"rank": r, "local_rank": r, "transport": "rccl", "nccl_comm_count": n_gpus, "nccl_user_rank": r, "nccl_device": r,
                              "reduction_comm_count": n_gpus, "hip_device": r, "pci_bus_id": f"0000:{0x05 + 16 * r:02x}:00.0"} for r in range(n_gpus)]
        full["comm_breakdown"] = _breakdown()
        full["transports_measured"] = {t: {"value": 30000.123456, "ms_per_step": 0.26, "timing": {"repeats": 6}, "comm_breakdown": _breakdown(),
                                           "postflight": {"ok": True}} for t in ("rccl", "rccl-plain", "ipc", "host")}
        full["transport_fallback"] = [{"transport": t, "reason": LONG, "seconds": 240.0} for t in ("rccl", "rccl-plain", "ipc")]
        full["contact_bicgstab"] = {"iterations": 10, "us_per_iteration": 500.0, "final_residual": 1.0, "comm_breakdown": _breakdown(), "error": LONG}
    return full


@pytest.mark.parametrize("n_gpus", [1, 2, 8])
def test_compact_line_fits_the_drivers_tail_and_is_strict_json(n_gpus):
    full = worst_case_record(n_gpus)
    line = bench.compact_line(full)
    assert len(line) <= bench.LINE_CAP == 8000 and "\n" not in line
    out = json.loads(line, parse_constant=lambda c: pytest.fail(f"non-strict JSON constant {c}"))
    for key in CONTRACT:
        assert key in out, key
    assert "dropped" not in out, out.get("dropped")  # the worst case fits WITHOUT dropping an optional block
    assert out["n_gpus"] == n_gpus and out["vs_baseline"] is None and out["dtype"] == "f64" and out["scaling"] == "weak"
    assert "workload" in out["config"] and "model" not in out["config"]
    r = out["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and r["kernel"] == "cg_step_march_kernel"
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-5 and r["traffic"] > 0
    # the SURVEY-8d-comparable figure sits INSIDE roofline: the stand-alone fp64-record SpMV
    assert r["spmv_general_frac_8d_rotating"] == pytest.approx(0.792478, rel=1e-5) and r["spmv_general_ms"] == pytest.approx(0.253302, rel=1e-5)
    assert r["spmv_general_bytes_8d"] == 1605894144
    c = out["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and len(c["sample"]) <= 200
    assert c["value_native_O3"] == 4.1 and c["value_native_O3_fast_math"] == 4.4
    assert out["value_general"] == pytest.approx(2278.18, rel=1e-5)
    assert out["configs"]["config4_gmres30_convdiff128"]["us_per_inner_iteration"] == pytest.approx(77.7488)
    assert len(out["configs"]["config3_bicgstab256"]["error"]) <= 100
    if n_gpus > 1:
        assert out["rccl"]["nccl_comm_count"] == n_gpus and out["rccl"]["distinct_devices"] == n_gpus
        assert len(out["rccl"]["ranks"]) == n_gpus and out["rccl"]["ranks"][1][:3] == [1, 1, 1]
        assert set(out["transports_measured"]) == {"rccl", "rccl-plain", "ipc", "host"}
        assert out["transports_measured"]["rccl"]["comm_breakdown"]["sendrecv_us_each"] == pytest.approx(80.48, rel=1e-3)
        assert out["preflight_ok"] is True and out["postflight_ok"] is True and len(out["transport_fallback"]) == 3


def test_compact_line_drops_optional_blocks_rather_than_exceed_the_cap():
    """Should a record outgrow the cap all the same, optional blocks go (and are named); the contract's fields stay."""
    line = bench.compact_line(worst_case_record(8), cap=3600)
    out = json.loads(line)
    assert len(line) <= 3600 and out["dropped"] and all(k in out for k in CONTRACT)


def test_compact_line_survives_missing_blocks_and_non_finite_numbers():
    full = {"metric": "m", "value": 1.0, "unit": "iter/s", "n_gpus": 1, "steps": 2, "warmup": 0, "ms_per_step": float("nan"),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": "w"},
            "roofline": {"kernel": "k", "bound": "hbm", "achieved": float("inf"), "peak": 8000.0, "unit": "GB/s", "frac": 0.5, "traffic": None},
            "cpu_baseline": {"error": "RuntimeError('no oracle')"}, "spmv": {"error": "boom"}, "blas1": {"error": "boom"}}
    out = json.loads(bench.compact_line(full))
    assert out["ms_per_step"] is None and out["roofline"]["achieved"] is None and out["roofline"]["traffic"] is None
    assert out["cpu_baseline"]["error"].startswith("RuntimeError")


def test_emit_writes_the_detail_file_and_prints_one_stdout_line(tmp_path, capsys):
    full = worst_case_record(2)
    path = tmp_path / "detail.json"
    bench.emit(full, str(path))
    cap = capsys.readouterr()
    lines = cap.out.splitlines()
    assert len(lines) == 1 and len(lines[0]) <= 8000 and json.loads(lines[0])["detail"] == str(path)
    detail = json.load(open(path))
    assert detail["spmv"]["general"]["rotating_3_pairs"]["launches"] == 60 and detail["roofline"]["note"] == LONG
    assert cap.err.startswith("bench_detail {")
