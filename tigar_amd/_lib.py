"""
ctypes binding of libtigar_hip.so (C-ABI declared in include/tigar_hip.h).

The product path has NO CPU fallback: if the shared library is missing, or no MI355X is
visible when a device call is made, this raises.  ``load(require_device=False)`` only loads
the library and declares prototypes (used by the CPU test-suite to check the exported
symbols); nothing is computed without a GPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TIGAR_LIB_PATH", os.path.join(_HERE, "libtigar_hip.so"))   # (override: A/B of two builds)

c_i32p = C.POINTER(C.c_int32)
c_i64p = C.POINTER(C.c_int64)
c_f64p = C.POINTER(C.c_double)
handle = C.c_void_p


class TigarHipError(RuntimeError):
    pass


class tg_dir_t(C.Structure):
    _fields_ = [("p", C.c_int32), ("nknots", C.c_int32), ("ghost", c_f64p),
                ("mult_first", C.c_int32), ("mult_last", C.c_int32), ("ncp", C.c_int32),
                ("nnodes", C.c_int64), ("nodes", c_f64p)]


class tg_kron1d_t(C.Structure):
    _fields_ = [("n", C.c_int64), ("m", C.c_int64), ("rowptr", c_i32p), ("col", c_i32p), ("val", c_f64p),
                ("t_rowptr", c_i32p), ("t_col", c_i32p), ("t_val", c_f64p)]


class tg_tensor_dir_t(C.Structure):
    _fields_ = [("p", C.c_int), ("nel", C.c_int), ("wl", c_f64p)]


class tg_tensor_pair_dir_t(C.Structure):
    _fields_ = [("p", C.c_int), ("nel", C.c_int), ("pr", C.c_int), ("pc", C.c_int), ("wlr", c_f64p), ("wlc", c_f64p)]


class tg_kron_dir_t(C.Structure):
    _fields_ = [("n", C.c_int64), ("rowptr", c_i32p), ("col", c_i32p), ("val", c_f64p)]


class tg_patch_t(C.Structure):
    _fields_ = [("d", C.c_int), ("p", C.c_int), ("verts", c_f64p * 3), ("nverts", C.c_int * 3),
                ("nsd", C.c_int), ("cp", handle * 4), ("nq", C.c_int)]


HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, c_f64p, C.c_int)
HOST_SENDRECV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, c_f64p, C.c_int64, c_f64p, C.c_int64)

# name -> (restype, argtypes); mirrors include/tigar_hip.h one to one
PROTOTYPES = {
    "tg_init": (C.c_int, [C.c_int]),
    "tg_shutdown": (C.c_int, []),
    "tg_last_error": (C.c_char_p, []),
    "tg_sync": (C.c_int, []),
    "tg_stream_set": (C.c_int, [C.c_int]),
    "tg_stream_wait": (C.c_int, [C.c_int, C.c_int]),
    "tg_device_info": (C.c_int, [C.c_char_p, C.c_int, C.POINTER(C.c_int), c_i64p]),
    "tg_mem_info": (C.c_int, [c_i64p, c_i64p]),
    "tg_pool_trim": (C.c_int, []),
    "tg_pool_stats": (C.c_int, [c_i64p, c_i64p, c_i64p]),
    "tg_timer_start": (C.c_int, [C.c_int]),
    "tg_timer_stop": (C.c_int, [C.c_int, c_f64p]),
    "tg_prof_reset": (C.c_int, []),
    "tg_prof_get": (C.c_int, [C.c_int, c_f64p, c_i64p]),
    "tg_vec_create": (C.c_int, [C.c_int64, C.POINTER(handle)]),
    "tg_vec_create_uninit": (C.c_int, [C.c_int64, C.POINTER(handle)]),
    "tg_vec_destroy": (C.c_int, [handle]),
    "tg_vec_size": (C.c_int, [handle, c_i64p]),
    "tg_vec_upload": (C.c_int, [handle, c_f64p, C.c_int64]),
    "tg_vec_download": (C.c_int, [handle, c_f64p, C.c_int64]),
    "tg_vec_fill": (C.c_int, [handle, C.c_double]),
    "tg_vec_copy": (C.c_int, [handle, handle]),
    "tg_vec_copy_range": (C.c_int, [handle, C.c_int64, handle, C.c_int64, C.c_int64]),
    "tg_vec_axpy": (C.c_int, [handle, C.c_double, handle]),
    "tg_vec_dot": (C.c_int, [handle, handle, c_f64p]),
    "tg_vec_norm": (C.c_int, [handle, C.c_int, c_f64p]),
    "tg_vec_zero_entries": (C.c_int, [handle, c_i32p, C.c_int64]),
    "tg_vec_zero_entries_offset": (C.c_int, [handle, c_i32p, C.c_int64, C.c_int64]),
    "tg_vec_tensor3": (C.c_int, [handle, C.c_int, C.POINTER(c_f64p), c_i64p, C.c_double,
                                 C.c_int64, C.c_int64]),
    "tg_csr_from_host": (C.c_int, [C.c_int64, C.c_int64, c_i64p, c_i32p, c_f64p, C.POINTER(handle)]),
    "tg_csr_dims": (C.c_int, [handle, c_i64p, c_i64p, c_i64p]),
    "tg_csr_download": (C.c_int, [handle, c_i64p, c_i32p, c_f64p]),
    "tg_csr_download_rows": (C.c_int, [handle, C.c_int64, C.c_int64, c_i64p, c_i32p, c_f64p, C.c_int64]),
    "tg_csr_destroy": (C.c_int, [handle]),
    "tg_csr_transpose": (C.c_int, [handle, C.POINTER(handle)]),
    "tg_csr_add": (C.c_int, [handle, handle, C.POINTER(handle)]),
    "tg_tensor_split": (C.c_int, [handle, handle, C.POINTER(handle), C.POINTER(handle)]),
    "tg_csr_block": (C.c_int, [handle, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.POINTER(handle)]),
    "tg_csr_select_columns": (C.c_int, [handle, C.POINTER(C.c_uint8), C.POINTER(handle)]),
    "tg_cellplan_ptap_extras": (C.c_int, [handle, handle, C.POINTER(handle), C.POINTER(handle)]),
    "tg_csr_nonempty_rows": (C.c_int, [handle, C.c_int64, c_i64p, C.POINTER(C.c_int64)]),
    "tg_csr_split_cells": (C.c_int, [handle, C.c_int, C.POINTER(handle), C.POINTER(handle)]),
    "tg_csr_from_blocks": (C.c_int, [C.c_int, C.POINTER(handle), C.POINTER(handle)]),
    "tg_csr_gather_rows": (C.c_int, [handle, c_i64p, C.c_int64, C.POINTER(handle)]),
    "tg_partition_mode": (C.c_int, [handle, c_i32p, C.c_int, c_i32p]),
    "tg_csr_permute_columns": (C.c_int, [handle, c_i32p, C.POINTER(handle)]),
    "tg_csr_from_triplets": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, c_i64p, c_i32p, c_f64p,
                                       C.c_double, C.POINTER(handle)]),
    "tg_extract_csr_tensor": (C.c_int, [C.c_int, C.POINTER(tg_dir_t), C.c_int32, C.c_int64,
                                        C.c_double, C.c_int64, C.c_int64, C.POINTER(handle)]),
    "tg_extract_csr_tensor_t": (C.c_int, [C.c_int, C.POINTER(tg_dir_t), C.c_int64, C.c_int64, C.c_double,
                                          C.c_int64, C.c_int64, C.POINTER(handle)]),
    "tg_extract_apply_tensor": (C.c_int, [C.c_int, C.POINTER(tg_dir_t), C.c_int32, C.c_double, C.c_int64,
                                          C.c_int64, handle, C.c_int64, handle]),
    "tg_extract_csr_points": (C.c_int, [C.c_int, C.POINTER(tg_dir_t), C.c_int32, C.c_int64,
                                        C.c_double, c_f64p, C.c_int64, C.POINTER(handle)]),
    "tg_extract_csr_bezier": (C.c_int, [C.c_int64, C.c_int, C.c_int, c_f64p, c_i64p, c_i32p, c_f64p, C.c_int32, C.c_int64,
                                        C.c_double, C.POINTER(handle)]),
    "tg_csr_vstack": (C.c_int, [C.c_int, C.POINTER(handle), C.POINTER(handle)]),
    "tg_csr_builder_create": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.POINTER(handle)]),
    "tg_csr_vstack_view": (C.c_int, [C.c_int, C.POINTER(handle), C.POINTER(handle)]),
    "tg_csr_builder_append": (C.c_int, [handle, handle]),
    "tg_csr_builder_finish": (C.c_int, [handle, C.POINTER(handle)]),
    "tg_csr_builder_destroy": (C.c_int, [handle]),
    "tg_eval_basis_1d": (C.c_int, [C.POINTER(tg_dir_t), c_f64p, C.c_int64, c_i32p, c_i32p, c_f64p]),
    "tg_spmv": (C.c_int, [handle, handle, handle]),
    "tg_spmv_offset": (C.c_int, [handle, handle, C.c_int64, handle]),
    "tg_spmv_sell": (C.c_int, [handle, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int64)]),
    "tg_spmv_symgrid": (C.c_int, [handle, C.c_int64, handle, handle, C.POINTER(C.c_int), C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "tg_spmm_host": (C.c_int, [handle, c_f64p, C.c_int, c_f64p]),
    "tg_spmv_t": (C.c_int, [handle, handle, handle]),
    "tg_ptap_symbolic": (C.c_int, [handle, C.c_int64, handle, C.c_int64, handle, C.c_int64,
                                   C.POINTER(handle)]),
    "tg_ptap_numeric": (C.c_int, [handle, handle, handle, handle, c_i32p, C.c_int64, C.c_double,
                                  C.POINTER(handle)]),
    "tg_ptap_destroy": (C.c_int, [handle]),
    "tg_ptap_prefer": (C.c_int, [C.c_int]),
    "tg_cellplan_create": (C.c_int, [C.c_int64, C.c_int, C.c_int, C.c_int64, c_f64p, c_i32p, c_i32p, handle, C.c_int, C.c_double,
                                     C.POINTER(handle)]),
    "tg_cellplan_ptap": (C.c_int, [handle, handle, c_i32p, C.c_int64, C.c_double, C.POINTER(handle)]),
    "tg_cellplan_destroy": (C.c_int, [handle]),
    "tg_cells_from_host": (C.c_int, [c_i32p, C.c_int64, C.c_int, C.POINTER(handle)]),
    "tg_cells_from_grid": (C.c_int, [C.c_int, c_i64p, C.c_int, c_i64p, c_i64p, C.POINTER(handle)]),
    "tg_cells_dims": (C.c_int, [handle, c_i64p, C.POINTER(C.c_int)]),
    "tg_cells_download": (C.c_int, [handle, c_i32p]),
    "tg_cells_destroy": (C.c_int, [handle]),
    "tg_elemplan_create": (C.c_int, [handle, C.c_int64, C.c_int64, handle, C.c_int64, C.POINTER(handle)]),
    "tg_elemplan_info": (C.c_int, [handle, c_i64p, c_i64p, C.POINTER(C.c_int), C.POINTER(C.c_int), c_i64p]),
    "tg_elemplan_ptap": (C.c_int, [handle, handle, C.c_int64, C.c_int64, C.c_int64, c_i32p, C.c_int64, C.c_double,
                                   C.POINTER(handle)]),
    "tg_elemplan_destroy": (C.c_int, [handle]),
    "tg_foldplan_create": (C.c_int, [handle, C.c_int64, handle, handle, C.c_int64, handle, C.POINTER(handle)]),
    "tg_foldplan_apply": (C.c_int, [handle, handle, c_i32p, C.c_int64, C.c_double, C.POINTER(handle)]),
    "tg_foldplan_destroy": (C.c_int, [handle]),
    "tg_ptap_kron": (C.c_int, [handle, C.c_int64, C.c_int, c_i64p, C.POINTER(tg_kron1d_t), C.c_int64, C.c_int64,
                               c_i32p, C.c_int64, C.c_double, C.POINTER(handle)]),
    "tg_ptap_kron_stage": (C.c_int, [handle, C.c_int64, C.c_int, c_i64p, C.POINTER(tg_kron1d_t), C.c_int64, C.c_int64,
                                     C.POINTER(handle)]),
    "tg_ptap_kron_append": (C.c_int, [handle, C.c_int64, C.c_int, c_i64p, C.POINTER(tg_kron1d_t), C.c_int64, C.c_int64,
                                      c_i32p, C.c_int64, C.c_double, handle]),
    "tg_tensor_plan_create": (C.c_int, [C.c_int, C.POINTER(tg_tensor_dir_t), C.POINTER(handle)]),
    "tg_tensor_plan_create_pair": (C.c_int, [C.c_int, C.POINTER(tg_tensor_pair_dir_t), C.POINTER(handle)]),
    "tg_tensor_plan_destroy": (C.c_int, [handle]),
    "tg_tensor2_plan_create": (C.c_int, [C.c_int, C.POINTER(tg_tensor_dir_t), C.POINTER(handle)]),
    "tg_tensor2_plan_create_pair": (C.c_int, [C.POINTER(tg_tensor_pair_dir_t), C.POINTER(handle)]),
    "tg_tensor2_ptap": (C.c_int, [handle, handle, c_i32p, C.c_int64, C.c_double, C.POINTER(handle)]),
    "tg_tensor_planes": (C.c_int, [handle, handle, C.c_int64, C.c_int, C.c_int, C.POINTER(handle)]),
    "tg_tensor_planes_destroy": (C.c_int, [handle]),
    "tg_tensor_planes_kron": (C.c_int, [handle, C.c_int, C.POINTER(tg_kron_dir_t), C.c_int, C.c_int, C.POINTER(handle)]),
    "tg_tensor_zstage": (C.c_int, [handle, C.c_int, C.POINTER(handle), C.c_int, C.c_int, c_i32p, C.c_int64,
                                   C.c_double, handle, C.POINTER(handle)]),
    "tg_csr_compact": (C.c_int, [handle, C.POINTER(handle)]),
    "tg_csr_is_loose": (C.c_int, [handle, C.POINTER(C.c_int)]),
    "tg_csr_rowptr_at": (C.c_int, [handle, C.c_int64, c_i64p]),
    "tg_zero_rows_cols": (C.c_int, [handle, C.c_int64, c_i32p, C.c_int64, C.c_double]),
    "tg_krylov_solve": (C.c_int, [handle, handle, handle, C.c_int, C.c_int, C.c_double, C.c_double,
                                  C.c_int, C.c_int, handle, C.POINTER(C.c_int), c_f64p,
                                  C.POINTER(C.c_int)]),
    "tg_krylov_solve_flags": (C.c_int, [handle, handle, handle, C.c_int, C.c_int, C.c_double, C.c_double,
                                        C.c_int, C.c_int, C.c_int, handle, C.POINTER(C.c_int), c_f64p,
                                        C.POINTER(C.c_int)]),
    "tg_lu_band_info": (C.c_int, [handle, C.POINTER(C.c_int), C.POINTER(C.c_int), c_i64p]),
    "tg_lu_solve": (C.c_int, [handle, handle, handle, C.POINTER(C.c_int)]),
    "tg_chol_solve": (C.c_int, [handle, handle, handle, C.POINTER(C.c_int)]),
    "tg_kron_sum_csr": (C.c_int, [C.c_int, C.c_int, C.POINTER(tg_kron_dir_t), C.c_int64, C.c_int64,
                                  C.POINTER(handle)]),
    "tg_kron_csr_rect": (C.c_int, [C.c_int, C.c_int, C.POINTER(tg_kron_dir_t), c_i64p, C.c_int64, C.c_int64,
                                   C.c_int, C.c_double, C.c_int64, C.c_int64, C.POINTER(handle)]),
    "tg_kron3_csr": (C.c_int, [C.c_int, C.POINTER(tg_kron_dir_t), c_i64p, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                               C.POINTER(handle)]),
    "tg_vec_pointwise_mult": (C.c_int, [handle, handle, handle]),
    "tg_csr_combine": (C.c_int, [C.c_double, handle, C.c_double, handle, handle, C.POINTER(handle)]),
    "tg_tensor_apply_1d": (C.c_int, [C.c_int, c_i64p, C.c_int, C.c_int64, c_i32p, c_i32p, c_f64p, C.c_int64, handle, handle]),
    "tg_assemble_mapped_matrix": (C.c_int, [C.POINTER(tg_patch_t), C.c_int, C.POINTER(handle)]),
    "tg_assemble_mapped_load": (C.c_int, [C.POINTER(tg_patch_t), handle, handle]),
    "tg_assemble_mapped_matrix_rows": (C.c_int, [C.POINTER(tg_patch_t), C.c_int, C.c_int64, C.c_int64, C.c_int64,
                                                  C.POINTER(handle)]),
    "tg_assemble_mapped_load_rows": (C.c_int, [C.POINTER(tg_patch_t), handle, C.c_int64, C.c_int64, C.c_int64, handle]),
    "tg_assemble_mapped_elasticity_rows": (C.c_int, [C.POINTER(tg_patch_t), C.c_int, C.c_int, C.c_double, C.c_double,
                                                      C.c_int64, C.c_int64, C.c_int64, C.POINTER(handle)]),
    "tg_comm_unique_id": (C.c_int, [C.c_char_p]),
    "tg_comm_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(handle)]),
    "tg_comm_create2": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.POINTER(handle)]),
    "tg_comm_ipc_shm_bytes": (C.c_int, [C.POINTER(C.c_int64)]),
    "tg_comm_create_ipc": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(handle)]),
    "tg_comm_rank_device": (C.c_int, [handle, C.c_int, C.POINTER(C.c_int)]),
    "tg_comm_create_host": (C.c_int, [C.c_int, C.c_int, HOST_ALLREDUCE_FN, HOST_SENDRECV_FN, C.c_void_p,
                                      C.POINTER(handle)]),
    "tg_comm_info": (C.c_int, [handle, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "tg_device_count": (C.c_int, [C.POINTER(C.c_int)]),
    "tg_comm_set_slab": (C.c_int, [handle, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64]),
    "tg_comm_allreduce_sum": (C.c_int, [handle, c_f64p, C.c_int]),
    "tg_comm_halo_extend": (C.c_int, [handle, handle, handle]),
    "tg_comm_destroy": (C.c_int, [handle]),
    "tg_comm_selftest": (C.c_int, [handle, C.c_double]),
}

_lib = None
_device_ready = False


def load(require_device=True, device=None):
    """Loads libtigar_hip.so; with require_device also binds the GPU (tg_init)."""
    global _lib, _device_ready
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise TigarHipError(
                "libtigar_hip.so not found at %s -- build it with "
                "`python -c 'import __graft_entry__ as g; g.build()'` (there is no CPU fallback)"
                % LIB_PATH)
        # the host driver of the target boxes shares device memory between processes through dmabuf only; RCCL's
        # intra-node transport needs this set before the HIP runtime starts (kept if the caller chose a value)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)      # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    if require_device and not _device_ready:
        if device is None:
            if "TIGAR_DEVICE" in os.environ:
                device = int(os.environ["TIGAR_DEVICE"])
            else:
                # one process per GPU; more local ranks than GPUs share devices round-robin (the communicator
                # is then host-staged, tigar_amd/launch.py)
                ndev = C.c_int(0)
                if _lib.tg_device_count(C.byref(ndev)) != 0 or ndev.value < 1:
                    raise TigarHipError("no MI355X visible: %s -- there is no CPU fallback" % _lib.tg_last_error().decode())
                device = int(os.environ.get("LOCAL_RANK", "0")) % ndev.value
        rc = _lib.tg_init(int(device))
        if rc != 0:
            raise TigarHipError("tg_init(%d) failed: %s -- the extraction path needs an MI355X; "
                                "there is no CPU fallback" % (device, _lib.tg_last_error().decode()))
        _device_ready = True
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = _lib.tg_last_error().decode() if _lib is not None else "library not loaded"
        raise TigarHipError("%s failed (status %d): %s" % (what or "libtigar_hip call", rc, msg))


def lib():
    return load(True)
