"""ctypes binding of libstorm_hip.so -- the C ABI declared in include/storm_hip.h.

There is no CPU fallback: if the HIP library is missing or fails to load, importing this
module raises.  (``torch`` is imported first, when present, so that the process uses one HIP
runtime -- torch's bundled libamdhip64.so.7 has the same SONAME as ROCm's.)
"""
from __future__ import annotations

import ctypes as C
import os

try:  # plumbing only: fixes the HIP runtime load order, provides torch.distributed
    import torch  # noqa: F401
except Exception:  # pragma: no cover - torch is optional for single-GPU use
    torch = None

_HERE = os.path.dirname(os.path.abspath(__file__))
# (STORM_HIP_LIB: another build of the same library, for A/B runs of two source states on one box -- tools/)
LIB_PATH = os.environ.get("STORM_HIP_LIB") or os.path.join(_HERE, "libstorm_hip.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
        "(or `make -C stormruler_amd/csrc`).  stormruler_amd has no CPU fallback.")

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)

f64p = C.POINTER(C.c_double)
i64p = C.POINTER(C.c_int64)
i32p = C.POINTER(C.c_int32)
vp = C.c_void_p


class SolverParams(C.Structure):
    _fields_ = [("num_iterations", C.c_int64), ("absolute_error_tolerance", C.c_double),
                ("relative_error_tolerance", C.c_double), ("num_inner_iterations", C.c_int64),
                ("check_lag", C.c_int32), ("gram_schmidt", C.c_int32)]


class SolverResult(C.Structure):
    _fields_ = [("iterations", C.c_int64), ("absolute_error", C.c_double), ("relative_error", C.c_double),
                ("initial_error", C.c_double), ("converged", C.c_int32), ("path_fallback", C.c_int32),
                ("num_applies", C.c_int64)]


class OpStats(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("n_rows", "n_cols", "nnz_offdiag", "ell_slots", "tail_nnz",
                                          "tail_rows", "n_slices", "max_row_len", "n_interior_slices",
                                          "device_bytes", "record_bytes", "value_dictionary_size",
                                          "offset_dictionary_size", "paired_rows", "tiled_planes", "spmv_blocks", "xcd_run_blocks")]


class MeshView(C.Structure):
    _fields_ = [("dim", C.c_int32), ("n_nbrs", C.c_int32), ("n_cells", C.c_int64), ("n_halo", C.c_int64),
                ("n_faces", C.c_int64), ("n_bfaces", C.c_int64), ("inner", i64p), ("outer", i64p), ("area", f64p),
                ("center", f64p), ("volume", f64p), ("b_cell", i64p), ("b_area", f64p), ("b_center", f64p),
                ("global_id", i64p), ("halo_owner", i32p), ("nbr_rank", i32p), ("send_ptr", i64p), ("send_idx", i64p),
                ("recv_ptr", i64p)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                          C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double))

APPLY_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)  # storm_hip_apply_fn(user, y, x)

# name -> (restype, argtypes); every symbol include/storm_hip.h declares.
SIGNATURES = {
    "storm_hip_abi_version": (C.c_int, []),
    "storm_hip_last_error": (C.c_char_p, []),
    "storm_hip_ctx_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
    "storm_hip_ctx_destroy": (C.c_int, [vp]),
    "storm_hip_ctx_sync": (C.c_int, [vp]),
    "storm_hip_ctx_info": (C.c_int, [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), i64p]),
    "storm_hip_ctx_set_option": (C.c_int, [vp, C.c_char_p, C.c_int64]),
    "storm_hip_order_cells": (C.c_int, [C.c_int32, C.c_int64, C.POINTER(C.c_double), C.c_int32, C.POINTER(C.c_int64),
                                        C.POINTER(C.c_int32)]),
    "storm_hip_ctx_get_counter": (C.c_int, [vp, C.c_char_p, C.POINTER(C.c_int64)]),
    "storm_hip_ctx_get_spmv_profile": (C.c_int, [vp, i64p, f64p, f64p]),
    "storm_hip_ctx_get_spmv_profile_samples": (C.c_int, [vp, C.POINTER(C.c_double), C.c_int64, C.POINTER(C.c_int64)]),
    "storm_hip_timer_start": (C.c_int, [vp]),
    "storm_hip_timer_stop": (C.c_int, [vp, C.POINTER(C.c_float)]),
    "storm_hip_comm_unique_id": (C.c_int, [vp]),
    "storm_hip_ctx_comm_init": (C.c_int, [vp, vp, C.c_int, C.c_int]),
    "storm_hip_ctx_comm_init_host": (C.c_int, [vp, C.c_int, C.c_int, ALLREDUCE_FN, EXCHANGE_FN, vp]),
    "storm_hip_map": (C.c_int, [vp, vp, vp, C.POINTER(C.c_int32), C.c_int, C.POINTER(C.c_double), C.c_int]),
    "storm_hip_ctx_comm_size": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "storm_hip_ctx_comm_rccl_view": (C.c_int, [vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                               C.POINTER(C.c_int), C.c_char_p, C.c_int]),
    "storm_hip_ctx_comm_ipc_export": (C.c_int, [vp, C.c_int, C.c_int, C.c_int64, vp]),
    "storm_hip_ctx_comm_init_ipc": (C.c_int, [vp, vp]),
    "storm_hip_vec_create": (C.c_int, [vp, C.c_int64, C.c_int64, C.POINTER(vp)]),
    "storm_hip_vec_create_like": (C.c_int, [vp, C.POINTER(vp)]),
    "storm_hip_vec_destroy": (C.c_int, [vp]),
    "storm_hip_vec_size": (C.c_int, [vp, i64p, i64p]),
    "storm_hip_vec_upload": (C.c_int, [vp, f64p, C.c_int64]),
    "storm_hip_vec_download": (C.c_int, [vp, f64p, C.c_int64]),
    "storm_hip_vec_device_ptr": (C.c_int, [vp, C.POINTER(vp)]),
    "storm_hip_vec_context": (C.c_int, [vp, C.POINTER(vp)]),
    "storm_hip_vec_get": (C.c_int, [vp, C.c_int64, f64p]),
    "storm_hip_fill": (C.c_int, [vp, C.c_double]),
    "storm_hip_copy": (C.c_int, [vp, vp]),
    "storm_hip_scale": (C.c_int, [vp, C.c_double]),
    "storm_hip_div_scalar": (C.c_int, [vp, C.c_double]),
    "storm_hip_axpy": (C.c_int, [vp, C.c_double, vp]),
    "storm_hip_xpay": (C.c_int, [vp, vp, C.c_double]),
    "storm_hip_axpbz": (C.c_int, [vp, C.c_double, vp, C.c_double, vp]),
    "storm_hip_bicgstab_p": (C.c_int, [vp, vp, C.c_double, C.c_double, vp]),
    "storm_hip_fill_randomly": (C.c_int, [vp]),
    "storm_hip_rng_reset": (None, []),
    "storm_hip_lin3": (C.c_int, [vp, vp, C.c_double, C.c_double, vp, C.c_double, vp]),
    "storm_hip_vmul_add": (C.c_int, [vp, C.c_double, vp, vp]),
    "storm_hip_vdiv": (C.c_int, [vp, C.c_double, vp, vp]),
    "storm_hip_vmul": (C.c_int, [vp, vp, vp]),
    "storm_hip_dot": (C.c_int, [vp, vp, f64p]),
    "storm_hip_norm2": (C.c_int, [vp, f64p]),
    "storm_hip_multi_dot": (C.c_int, [vp, C.POINTER(vp), C.c_int, f64p]),
    "storm_hip_multi_dot_begin": (C.c_int, [vp, C.POINTER(vp), C.c_int, C.POINTER(C.c_int)]),
    "storm_hip_multi_dot_end": (C.c_int, [vp, C.c_int, f64p]),
    "storm_hip_multi_axpy": (C.c_int, [vp, f64p, C.POINTER(vp), C.c_int]),
    "storm_hip_op_create_from_faces": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int64, i64p, i64p, f64p,
                                                 C.c_int64, i64p, f64p, f64p, C.POINTER(vp)]),
    "storm_hip_op_create_from_mesh": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int32, C.c_int64, i64p, i64p, f64p, f64p,
                                                C.c_int64, i64p, f64p, f64p, f64p, C.POINTER(vp)]),
    "storm_hip_op_create_from_face_weights": (C.c_int, [vp, C.c_int64, C.c_int64, C.c_int64, i64p, i64p,
                                                        f64p, f64p, f64p, C.POINTER(vp)]),
    "storm_hip_op_create_csr": (C.c_int, [vp, C.c_int64, C.c_int64, i64p, i64p, f64p, C.POINTER(vp)]),
    "storm_hip_op_set_halo": (C.c_int, [vp, C.c_int, i32p, i64p, i64p, i64p]),
    "storm_hip_mesh_read_tetgen": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(vp)]),
    "storm_hip_mesh_from_simplices": (C.c_int, [C.c_int32, C.c_int64, f64p, C.c_int64, i64p, i64p, C.c_int64, i64p, C.POINTER(vp)]),
    "storm_hip_mesh_write_tetgen": (C.c_int, [C.c_char_p, C.c_int32, C.c_int64, f64p, C.c_int64, i64p, i64p, C.c_int64, i64p]),
    "storm_hip_mesh_create": (C.c_int, [C.c_int32, C.c_int64, C.c_int64, C.c_int64, i64p, i64p, f64p, f64p, f64p, C.c_int64,
                                        i64p, f64p, f64p, i64p, i32p, C.POINTER(vp)]),
    "storm_hip_mesh_get_view": (C.c_int, [vp, C.POINTER(MeshView)]),
    "storm_hip_mesh_permute_cells": (C.c_int, [vp, i64p]),
    "storm_hip_partition_rcb": (C.c_int, [C.c_int32, C.c_int64, f64p, C.c_int32, i32p]),
    "storm_hip_partition_slabs": (C.c_int, [C.c_int32, C.c_int64, f64p, C.c_int32, C.c_int32, i32p]),
    "storm_hip_mesh_partition": (C.c_int, [vp, i32p, C.c_int32, C.c_int32, C.POINTER(vp)]),
    "storm_hip_mesh_halo_plan": (C.c_int, [vp, C.c_int32]),
    "storm_hip_op_create_from_mesh_object": (C.c_int, [vp, vp, C.POINTER(vp)]),
    "storm_hip_mesh_destroy": (C.c_int, [vp]),
    "storm_hip_op_apply": (C.c_int, [vp, C.c_double, C.c_double, vp, vp]),
    "storm_hip_op_apply_add": (C.c_int, [vp, C.c_double, vp, vp]),
    "storm_hip_op_get_diagonal": (C.c_int, [vp, C.c_double, C.c_double, C.c_int, vp]),
    "storm_hip_op_get_stats": (C.c_int, [vp, C.POINTER(OpStats)]),
    "storm_hip_op_destroy": (C.c_int, [vp]),
    "storm_hip_solver_params_default": (None, [C.POINTER(SolverParams)]),
    "storm_hip_solve_cg": (C.c_int, [vp, C.c_double, C.c_double, vp, vp, C.POINTER(SolverParams),
                                     C.POINTER(SolverResult), f64p]),
    "storm_hip_solve_bicgstab": (C.c_int, [vp, C.c_double, C.c_double, vp, vp, C.POINTER(SolverParams),
                                           C.POINTER(SolverResult), f64p]),
    "storm_hip_solve_gmres": (C.c_int, [vp, C.c_double, C.c_double, vp, vp, C.POINTER(SolverParams),
                                        C.POINTER(SolverResult), f64p]),
    "storm_hip_krylov_create": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
    "storm_hip_krylov_destroy": (C.c_int, [vp]),
    "storm_hip_krylov_set_operator": (C.c_int, [vp, vp, C.c_double, C.c_double]),
    "storm_hip_krylov_set_operator_fn": (C.c_int, [vp, APPLY_FN, vp]),
    "storm_hip_krylov_set_preconditioner_fn": (C.c_int, [vp, APPLY_FN, vp, C.c_int]),
    "storm_hip_krylov_set_preconditioner_diag": (C.c_int, [vp, vp, C.c_int]),
    "storm_hip_krylov_set_real": (C.c_int, [vp, C.c_char_p, C.c_double]),
    "storm_hip_krylov_solve": (C.c_int, [vp, vp, vp, C.POINTER(SolverParams), C.POINTER(SolverResult), f64p, i64p]),
    "storm_hip_krylov_init": (C.c_int, [vp, vp, vp, C.POINTER(SolverParams), f64p]),
    "storm_hip_krylov_iterate": (C.c_int, [vp, f64p]),
    "storm_hip_krylov_finalize": (C.c_int, [vp]),
}

for _name, (_res, _args) in SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here == the library does not export the symbol
    _fn.restype = _res
    _fn.argtypes = _args

if lib.storm_hip_abi_version() != 6:
    raise ImportError("libstorm_hip.so ABI version mismatch")


class StormHipError(RuntimeError):
    """A nonzero C-ABI status (the C++ adapter throws std::runtime_error the same way)."""

    def __init__(self, status: int, what: str):
        super().__init__(f"storm_hip status {status}: {what}")
        self.status = status


def check(status: int) -> None:
    if status != 0:
        raise StormHipError(status, (lib.storm_hip_last_error() or b"").decode(errors="replace"))
