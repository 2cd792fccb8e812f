"""ctypes binding of libcurdle_g1.so (C ABI: include/curdle_g1.h).

No torch, no fallbacks: if the shared library is missing this module raises at import, and every
device entry point raises `NativeError` when no GPU / HIP error -- there is no CPU path for the MSM.
"""
from __future__ import annotations

import ctypes
import os
import threading
from ctypes import POINTER, c_char_p, c_float, c_int, c_size_t, c_uint64, c_void_p

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CURDLE_G1_LIB") or os.path.join(HERE, "libcurdle_g1.so")

POINT_BYTES = 144
NPHASE = 7
PHASE_NAMES = ("prepare", "sort_count", "sort_scatter", "chunks", "accumulate", "seg_reduce", "bit_tree")

OK, ERR_ARG, ERR_HIP, ERR_ENCODING, ERR_NOT_ON_CURVE, ERR_NOT_IN_SUBGROUP, ERR_COMM = range(7)


class NativeError(RuntimeError):
    """HIP / argument failure inside libcurdle_g1.so (never silently replaced by a CPU path)."""


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `python -m curdleproofs_pie_amd.build` "
        "(hipcc --offload-arch=gfx950).  There is no pure-Python fallback."
    )

# The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4; read at its first call).  A verifier
# with the front-end on the device keeps seven or more streams busy (decoding, MSM, copies, several front-end launches): with four queues
# only two of its front-end launches ever ran side by side (profiles/r03_frontend_device_bench.txt), and with eight a 26 ms front-end
# launch still shared a queue with the decoding or MSM stream behind it: 13.5 ms per batch of 1024 proofs, against 7.8 ms with 24 queues
# (profiles/r03_verify_fe_ab.txt; the MSM benchmark itself is indifferent: 3.03 ms either way).  The library does NOT touch the host
# process's environment by itself: an application that wants the verifier's full throughput calls tune_runtime() (or exports the
# variable) before anything initialises HIP; ShuffleBatchVerifier sizes its pipelines by hw_queues() either way.
DEFAULT_HW_QUEUES = 4
TUNED_HW_QUEUES = 24


def hw_queues() -> int:
    """The hardware queues the HIP runtime of this process will use (GPU_MAX_HW_QUEUES, else the runtime's default of 4)."""
    try:
        return max(1, int(os.environ.get("GPU_MAX_HW_QUEUES", DEFAULT_HW_QUEUES)))
    except ValueError:
        return DEFAULT_HW_QUEUES


def tune_runtime(queues: int = TUNED_HW_QUEUES) -> int:
    """Opt-in: export GPU_MAX_HW_QUEUES = `queues` unless the variable is already set.  Only effective BEFORE the first HIP call of the
    process (creating a Context, torch.cuda, ...); bench.py, the tools and the test suite call it first thing.  Returns hw_queues()."""
    os.environ.setdefault("GPU_MAX_HW_QUEUES", str(int(queues)))
    return hw_queues()


lib = ctypes.CDLL(LIB_PATH)


def build_info() -> dict:
    """How the loaded library was built: {"pipeline": "staged" | "plain", "dropped_passes": [...]} (build.py)."""
    import json

    try:
        with open(LIB_PATH + ".buildinfo") as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


_u8p = c_char_p  # immutable byte buffers in
_buf = c_void_p  # mutable buffers out (ctypes.create_string_buffer / addresses)


def _proto(name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)
    return fn


# host single-element ops
cg1_identity = _proto("cg1_identity", None, _buf)
cg1_generator = _proto("cg1_generator", None, _buf)
cg1_add = _proto("cg1_add", None, _buf, _u8p, _u8p)
cg1_sub = _proto("cg1_sub", None, _buf, _u8p, _u8p)
cg1_neg = _proto("cg1_neg", None, _buf, _u8p)
cg1_double = _proto("cg1_double", None, _buf, _u8p)
cg1_mul = _proto("cg1_mul", None, _buf, _u8p, _u8p)
cg1_eq = _proto("cg1_eq", c_int, _u8p, _u8p)
cg1_is_identity = _proto("cg1_is_identity", c_int, _u8p)
cg1_compress = _proto("cg1_compress", None, _buf, _u8p)
cg1_decompress = _proto("cg1_decompress", c_int, _buf, _u8p, c_int)
cg1_to_affine96 = _proto("cg1_to_affine96", None, _buf, _u8p)
cg1_from_affine96 = _proto("cg1_from_affine96", c_int, _buf, _u8p, c_int)
cg1_batch_from_affine96 = _proto("cg1_batch_from_affine96", c_int, _buf, c_void_p, c_size_t)
cg1_batch_to_affine96 = _proto("cg1_batch_to_affine96", None, _buf, _u8p, c_size_t)
cg1_batch_decompress = _proto("cg1_batch_decompress", c_int, _buf, _u8p, c_size_t, c_int, POINTER(c_size_t))
cg1_batch_compress = _proto("cg1_batch_compress", None, _buf, _u8p, c_size_t)
# device
cg1_device_count = _proto("cg1_device_count", c_int)
cg1_ctx_create = _proto("cg1_ctx_create", c_void_p, c_int)
cg1_ctx_create_cu_mask = _proto("cg1_ctx_create_cu_mask", c_void_p, c_int, POINTER(ctypes.c_uint32), c_size_t)
cg1_ctx_destroy = _proto("cg1_ctx_destroy", None, c_void_p)
cg1_ctx_error = _proto("cg1_ctx_error", c_char_p, c_void_p)
cg1_dev_malloc = _proto("cg1_dev_malloc", c_void_p, c_void_p, c_size_t)
cg1_dev_free = _proto("cg1_dev_free", None, c_void_p, c_void_p)
cg1_h2d = _proto("cg1_h2d", c_int, c_void_p, c_void_p, c_void_p, c_size_t)
cg1_d2h = _proto("cg1_d2h", c_int, c_void_p, c_void_p, c_void_p, c_size_t)
cg1_host_alloc = _proto("cg1_host_alloc", c_void_p, c_void_p, c_size_t)
cg1_host_free = _proto("cg1_host_free", None, c_void_p, c_void_p)
cg1_h2d_async = _proto("cg1_h2d_async", c_int, c_void_p, c_void_p, c_void_p, c_size_t)
cg1_copy_fence = _proto("cg1_copy_fence", c_int, c_void_p)
cg1_stream_sync = _proto("cg1_stream_sync", c_int, c_void_p)
cg1_d2h_2d = _proto("cg1_d2h_2d", c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_size_t, c_size_t)
cg1_ctx_sync = _proto("cg1_ctx_sync", c_int, c_void_p)
cg1_ctx_set_param = _proto("cg1_ctx_set_param", c_int, c_void_p, c_char_p, c_int)
cg1_ctx_device = _proto("cg1_ctx_device", c_int, c_void_p)
cg1_ctx_stream = _proto("cg1_ctx_stream", c_void_p, c_void_p)
# multi-GPU: one process over a device list, and the per-rank communicator (TCP control channel + RCCL)
cg1_msm_multi_device = _proto("cg1_msm_multi_device", c_int, POINTER(c_void_p), c_size_t, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_size_t), c_int, _buf)
cg1_comm_create = _proto("cg1_comm_create", c_void_p, c_int, c_int)
cg1_comm_port = _proto("cg1_comm_port", c_int, c_void_p)
cg1_comm_rank = _proto("cg1_comm_rank", c_int, c_void_p)
cg1_comm_connect = _proto("cg1_comm_connect", c_int, c_void_p, c_char_p, c_int, c_uint64, c_int)
cg1_comm_set_timeout = _proto("cg1_comm_set_timeout", c_int, c_void_p, c_int)
cg1_comm_attach_rccl = _proto("cg1_comm_attach_rccl", c_int, c_void_p, c_void_p)
cg1_comm_transport = _proto("cg1_comm_transport", c_char_p, c_void_p)
cg1_comm_world_seen = _proto("cg1_comm_world_seen", c_int, c_void_p)
cg1_comm_error = _proto("cg1_comm_error", c_char_p, c_void_p)
cg1_comm_allgather = _proto("cg1_comm_allgather", c_int, c_void_p, c_void_p, c_size_t, c_void_p)
cg1_comm_allgather_host = _proto("cg1_comm_allgather_host", c_int, c_void_p, c_void_p, c_size_t, c_void_p)
cg1_comm_barrier = _proto("cg1_comm_barrier", c_int, c_void_p)
cg1_comm_allreduce_g1 = _proto("cg1_comm_allreduce_g1", c_int, c_void_p, _u8p, _buf, c_void_p)
cg1_comm_destroy = _proto("cg1_comm_destroy", None, c_void_p)
cg1_msm = _proto("cg1_msm", c_int, c_void_p, _u8p, _u8p, c_size_t, _buf)
cg1_msm_addr = lib["cg1_msm"]       # the same entry point taking raw addresses (page-locked staging) instead of bytes objects
cg1_msm_addr.restype = c_int
cg1_msm_addr.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t, _buf]
cg1_msm_device = _proto("cg1_msm_device", c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int, _buf)
cg1_msm_device_begin = _proto("cg1_msm_device_begin", c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_int, c_int)
cg1_msm_device_end = _proto("cg1_msm_device_end", c_int, c_void_p, _buf)
cg1_msm_blobs = _proto("cg1_msm_blobs", c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_int, _buf)
cg1_stage_reserve = _proto("cg1_stage_reserve", c_int, c_void_p, c_size_t, c_size_t, POINTER(c_void_p), POINTER(c_void_p))
cg1_msm_blobs_device = _proto("cg1_msm_blobs_device", c_int, c_void_p, c_void_p, c_void_p, c_size_t, c_int, _buf)
cg1_vec_create = _proto("cg1_vec_create", c_void_p, c_void_p, c_void_p, c_size_t, c_int)
cg1_vec_destroy = _proto("cg1_vec_destroy", None, c_void_p)
cg1_vec_len = _proto("cg1_vec_len", c_size_t, c_void_p)
cg1_msm_vec = _proto("cg1_msm_vec", c_int, c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, _buf)
cg1_batch_normalize = _proto("cg1_batch_normalize", c_int, c_void_p, c_size_t, c_void_p, c_void_p)
cg1_msm_batched_device = _proto("cg1_msm_batched_device", c_int, c_void_p, c_void_p, c_void_p, POINTER(ctypes.c_uint32), c_size_t, c_int, _buf)
cg1_msm_batched = _proto("cg1_msm_batched", c_int, c_void_p, _u8p, _u8p, POINTER(ctypes.c_uint32), c_size_t, _buf)
cg1_get_timings = _proto("cg1_get_timings", c_int, c_void_p, POINTER(c_float), POINTER(c_float), POINTER(c_int))
cg1_get_host_timings = _proto("cg1_get_host_timings", c_int, c_void_p, POINTER(c_float))
cg1_get_last_counts = _proto("cg1_get_last_counts", c_int, c_void_p, POINTER(ctypes.c_uint32), POINTER(ctypes.c_uint32))
cg1_get_last_launches = _proto("cg1_get_last_launches", c_int, c_void_p)
cg1_plan_describe = _proto("cg1_plan_describe", c_int, c_int, c_int, POINTER(c_int), POINTER(c_int), c_int)
cg1_timer_begin = _proto("cg1_timer_begin", c_int, c_void_p)
cg1_timer_end = _proto("cg1_timer_end", c_int, c_void_p, POINTER(c_float))
cg1_batch_mul_device = _proto("cg1_batch_mul_device", c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t)
cg1_batch_mul_add_device = _proto("cg1_batch_mul_add_device", c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_size_t)
cg1_batch_mul_add = _proto("cg1_batch_mul_add", c_int, c_void_p, _u8p, c_size_t, _u8p, c_size_t, _u8p, _buf, c_size_t)
cg1_batch_mul_add_pool = _proto("cg1_batch_mul_add_pool", c_int, _u8p, c_size_t, _u8p, c_size_t, _u8p, _buf, c_size_t, c_int)
cg1_batch_decompress_device = _proto("cg1_batch_decompress_device", c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int)
cg1_batch_decompress_enqueue = _proto("cg1_batch_decompress_enqueue", c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_int)
cg1_subgroup_flags_enqueue = _proto("cg1_subgroup_flags_enqueue", c_int, c_void_p, c_void_p, c_size_t, c_size_t, POINTER(ctypes.c_uint32), c_size_t, c_void_p)
cg1_side_sync = _proto("cg1_side_sync", c_int, c_void_p)
cg1_batch_compress_device = _proto("cg1_batch_compress_device", c_int, c_void_p, c_void_p, c_void_p, c_size_t)
cg1_batch_decompress_gpu = _proto("cg1_batch_decompress_gpu", c_int, c_void_p, _u8p, _buf, c_size_t, c_int, POINTER(c_size_t))
cg1_gen_scalars_device = _proto("cg1_gen_scalars_device", c_int, c_void_p, c_void_p, c_size_t, c_uint64)
cg1_probe_madd = _proto("cg1_probe_madd", c_int, c_void_p, c_void_p, c_size_t, c_size_t, c_int, POINTER(c_float))
cg1_probe_add_chain = _proto("cg1_probe_add_chain", c_int, c_void_p, c_int, c_void_p, c_size_t, c_int, c_int, c_void_p, POINTER(c_float))
cg1_probe_mad_rate = _proto("cg1_probe_mad_rate", c_int, c_void_p, c_int, c_int, POINTER(ctypes.c_double))
cg1_batch_sum_device = _proto("cg1_batch_sum_device", c_int, c_void_p, c_void_p, POINTER(ctypes.c_uint32), c_size_t, c_void_p)
cg1_batch_sum = _proto("cg1_batch_sum", c_int, c_void_p, _u8p, POINTER(ctypes.c_uint32), c_size_t, _buf)
# deferred evaluation of the G1Point operators (csrc/lazy_host.cpp, csrc/capi_lincomb.h)
cg1_validate_compressed = _proto("cg1_validate_compressed", c_int, _u8p, POINTER(c_int))
cg1_fp_jacobi = _proto("cg1_fp_jacobi", c_int, _u8p)
cg1_batch_decompress_pool = _proto("cg1_batch_decompress_pool", c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_int, POINTER(c_size_t))
cg1_batch_decompress_rows = _proto("cg1_batch_decompress_rows", c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, POINTER(c_size_t))
cg1_batch_subgroup_pool = _proto("cg1_batch_subgroup_pool", c_int, c_void_p, c_size_t, c_void_p, c_int)
cg1_batch_subgroup = _proto("cg1_batch_subgroup", c_int, c_void_p, c_void_p, c_size_t, c_void_p, POINTER(c_int))
cg1_lincomb_batch = _proto("cg1_lincomb_batch", c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, POINTER(c_int))
cg1_glv_split = _proto("cg1_glv_split", None, _u8p, c_void_p, c_void_p, POINTER(c_int), POINTER(c_int))
cg1_lincomb_batch_pool = _proto("cg1_lincomb_batch_pool", c_int, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int)

# native Merlin transcript (host)
MERLIN_STATE_BYTES = 208
cg1_keccak_f1600 = _proto("cg1_keccak_f1600", None, _buf)
cg1_keccak_f1600_x8 = _proto("cg1_keccak_f1600_x8", None, _buf)
cg1_keccak_f1600_x8_states = _proto("cg1_keccak_f1600_x8_states", None, c_void_p, c_int)
cg1_strobe_new = _proto("cg1_strobe_new", None, _buf, _u8p, c_size_t)
cg1_strobe_meta_ad = _proto("cg1_strobe_meta_ad", c_int, _buf, _u8p, c_size_t, c_int)
cg1_strobe_ad = _proto("cg1_strobe_ad", c_int, _buf, _u8p, c_size_t, c_int)
cg1_strobe_prf = _proto("cg1_strobe_prf", c_int, _buf, _buf, c_size_t, c_int)
cg1_strobe_key = _proto("cg1_strobe_key", c_int, _buf, _u8p, c_size_t, c_int)
cg1_merlin_init = _proto("cg1_merlin_init", None, _buf, _u8p, c_size_t)
cg1_merlin_append = _proto("cg1_merlin_append", None, _buf, _u8p, c_size_t, _u8p, c_size_t)
cg1_merlin_append_list = _proto("cg1_merlin_append_list", None, _buf, _u8p, c_size_t, _u8p, c_size_t, c_size_t)
cg1_merlin_challenge = _proto("cg1_merlin_challenge", None, _buf, _u8p, c_size_t, _buf, c_size_t)
cg1_merlin_challenge_scalar = _proto("cg1_merlin_challenge_scalar", None, _buf, _u8p, c_size_t, _buf)



class MerlinOp(ctypes.Structure):
    """cg1_merlin_op: one step of the batched device transcript (include/curdle_g1.h)."""
    _fields_ = [("kind", ctypes.c_uint8), ("label_len", ctypes.c_uint8), ("pad", ctypes.c_uint16), ("len", ctypes.c_uint32),
                ("data_off", ctypes.c_uint32), ("out_off", ctypes.c_uint32), ("label", ctypes.c_uint8 * 32)]


cg1_merlin_last_passes = _proto("cg1_merlin_last_passes", c_int, c_void_p)
cg1_merlin_last_kernel = _proto("cg1_merlin_last_kernel", c_int, c_void_p)
cg1_merlin_block_program_emulate = _proto("cg1_merlin_block_program_emulate", c_int, _u8p, c_void_p, c_size_t, _u8p, c_size_t, _buf, c_size_t, _buf, POINTER(ctypes.c_uint32))
cg1_merlin_batch_device = _proto("cg1_merlin_batch_device", c_int, c_void_p, c_void_p, POINTER(MerlinOp), c_size_t, c_void_p, c_size_t, c_void_p, c_size_t, c_void_p, c_size_t)

# batch verifier front-end of the shuffle argument (host)
cg1_shuffle_crs_create = _proto("cg1_shuffle_crs_create", c_void_p, _u8p, c_size_t, c_size_t)
cg1_shuffle_crs_destroy = _proto("cg1_shuffle_crs_destroy", None, c_void_p)
cg1_shuffle_proof_bytes = _proto("cg1_shuffle_proof_bytes", c_size_t, c_void_p)
cg1_shuffle_points_per_proof = _proto("cg1_shuffle_points_per_proof", c_size_t, c_void_p)
cg1_shuffle_crs_points = _proto("cg1_shuffle_crs_points", c_size_t, c_void_p)
cg1_shuffle_challenges_per_proof = _proto("cg1_shuffle_challenges_per_proof", c_size_t, c_void_p)
# (all pointer arguments c_void_p: bytes objects, ctypes buffers and raw addresses -- sub-ranges of a batch -- are accepted)
cg1_shuffle_prepare = _proto("cg1_shuffle_prepare", c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int)
cg1_opening_prepare = _proto("cg1_opening_prepare", c_int, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p)
cg1_shuffle_set_grouped = _proto("cg1_shuffle_set_grouped", None, c_int)
cg1_shuffle_default_threads = _proto("cg1_shuffle_default_threads", c_size_t)
cg1_shuffle_rowin_scalars = _proto("cg1_shuffle_rowin_scalars", c_size_t, c_void_p)
cg1_shuffle_prepare_inputs = _proto("cg1_shuffle_prepare_inputs", c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_int)
cg1_shuffle_rows_device = _proto("cg1_shuffle_rows_device", c_int, c_void_p, c_size_t, c_size_t, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p)
cg1_shuffle_fe_create = _proto("cg1_shuffle_fe_create", c_void_p, c_void_p, c_size_t, c_size_t, c_void_p, c_void_p)
cg1_shuffle_fe_destroy = _proto("cg1_shuffle_fe_destroy", None, c_void_p)
cg1_shuffle_fe_aux_bytes = _proto("cg1_shuffle_fe_aux_bytes", c_size_t)
cg1_shuffle_gather_aux = _proto("cg1_shuffle_gather_aux", c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p)
cg1_shuffle_fe_nodes = _proto("cg1_shuffle_fe_nodes", c_size_t, c_void_p)
cg1_shuffle_fe_last_passes = _proto("cg1_shuffle_fe_last_passes", c_size_t, c_void_p, c_void_p)
cg1_shuffle_fe_program_shape = _proto("cg1_shuffle_fe_program_shape", c_int, c_size_t, c_size_t, POINTER(ctypes.c_uint32))
cg1_shuffle_fe_emulate_to_first_barrier = _proto("cg1_shuffle_fe_emulate_to_first_barrier", c_int, c_size_t, c_size_t, _u8p, _u8p, _buf, c_size_t, POINTER(ctypes.c_uint32))
cg1_shuffle_fe_last_split = _proto("cg1_shuffle_fe_last_split", None, c_void_p, POINTER(ctypes.c_uint32))
cg1_shuffle_fe_enqueue = _proto("cg1_shuffle_fe_enqueue", c_int, c_void_p, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int)
cg1_shuffle_exact_same_scalar = _proto("cg1_shuffle_exact_same_scalar", c_int, c_void_p, c_void_p, c_void_p, POINTER(c_int))
cg1_opening_prepare_device = _proto("cg1_opening_prepare_device", c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p)
cg1_opening_weights_from_seed = _proto("cg1_opening_weights_from_seed", c_int, c_void_p, c_size_t, c_size_t, c_void_p)
cg1_opening_exact = _proto("cg1_opening_exact", c_int, c_void_p, c_void_p, c_void_p, POINTER(c_int))
cg1_opening_exact_status = _proto("cg1_opening_exact_status", c_int, c_void_p, c_void_p, c_void_p, POINTER(c_int))
cg1_shuffle_gather_points = _proto("cg1_shuffle_gather_points", c_int, c_void_p, c_size_t, c_void_p, c_void_p, c_void_p)
cg1_shuffle_apply_point_status = _proto("cg1_shuffle_apply_point_status", c_int, _buf, _u8p, c_size_t, c_size_t, _buf, _buf, c_size_t)
cg1_shuffle_sum_crs_scalars = _proto("cg1_shuffle_sum_crs_scalars", c_int, _buf, _buf, c_size_t, c_size_t, _buf)

EXPORTED_SYMBOLS = [
    "cg1_shuffle_crs_create", "cg1_shuffle_crs_destroy", "cg1_shuffle_proof_bytes", "cg1_shuffle_points_per_proof",
    "cg1_shuffle_crs_points", "cg1_shuffle_challenges_per_proof", "cg1_opening_prepare", "cg1_shuffle_prepare", "cg1_shuffle_set_grouped", "cg1_shuffle_default_threads", "cg1_shuffle_gather_points", "cg1_shuffle_rowin_scalars", "cg1_shuffle_prepare_inputs", "cg1_shuffle_rows_device", "cg1_shuffle_exact_same_scalar", "cg1_opening_exact", "cg1_opening_exact_status", "cg1_opening_prepare_device", "cg1_opening_weights_from_seed", "cg1_subgroup_flags_enqueue", "cg1_side_sync", "cg1_shuffle_apply_point_status", "cg1_shuffle_sum_crs_scalars",
    "cg1_keccak_f1600", "cg1_keccak_f1600_x8", "cg1_keccak_f1600_x8_states", "cg1_strobe_new", "cg1_strobe_meta_ad", "cg1_strobe_ad", "cg1_strobe_prf", "cg1_strobe_key", "cg1_merlin_init",
    "cg1_merlin_append", "cg1_merlin_append_list", "cg1_merlin_challenge", "cg1_merlin_challenge_scalar", "cg1_merlin_batch_device",
    "cg1_identity", "cg1_generator", "cg1_add", "cg1_sub", "cg1_neg", "cg1_double", "cg1_mul", "cg1_eq",
    "cg1_is_identity", "cg1_compress", "cg1_decompress", "cg1_to_affine96", "cg1_from_affine96",
    "cg1_batch_to_affine96", "cg1_batch_decompress", "cg1_batch_compress", "cg1_device_count", "cg1_ctx_create",
    "cg1_ctx_destroy", "cg1_ctx_error", "cg1_dev_malloc", "cg1_dev_free", "cg1_h2d", "cg1_d2h", "cg1_d2h_2d", "cg1_h2d_async", "cg1_copy_fence", "cg1_stream_sync", "cg1_batch_decompress_enqueue", "cg1_host_alloc", "cg1_host_free", "cg1_ctx_sync", "cg1_ctx_set_param",
    "cg1_msm", "cg1_msm_device", "cg1_msm_device_begin", "cg1_msm_device_end", "cg1_msm_batched_device", "cg1_msm_batched", "cg1_get_timings", "cg1_get_host_timings", "cg1_get_last_counts", "cg1_timer_begin", "cg1_timer_end", "cg1_batch_mul_device", "cg1_batch_mul_add_device", "cg1_batch_mul_add", "cg1_batch_mul_add_pool", "cg1_batch_decompress_device", "cg1_batch_compress_device", "cg1_batch_decompress_gpu", "cg1_gen_scalars_device", "cg1_probe_madd",
    "cg1_ctx_create_cu_mask", "cg1_shuffle_fe_create", "cg1_shuffle_fe_destroy", "cg1_shuffle_fe_aux_bytes", "cg1_shuffle_gather_aux", "cg1_shuffle_fe_enqueue", "cg1_shuffle_fe_nodes", "cg1_shuffle_fe_last_passes", "cg1_shuffle_fe_last_split", "cg1_shuffle_fe_program_shape", "cg1_shuffle_fe_emulate_to_first_barrier",
    "cg1_merlin_last_passes", "cg1_merlin_last_kernel", "cg1_merlin_block_program_emulate", "cg1_probe_mad_rate", "cg1_batch_sum_device", "cg1_batch_sum", "cg1_ctx_device", "cg1_ctx_stream", "cg1_msm_multi_device",
    "cg1_comm_create", "cg1_comm_port", "cg1_comm_rank", "cg1_comm_connect", "cg1_comm_set_timeout", "cg1_comm_attach_rccl", "cg1_comm_transport",
    "cg1_comm_world_seen", "cg1_comm_error", "cg1_comm_allgather", "cg1_comm_allgather_host", "cg1_comm_barrier", "cg1_comm_allreduce_g1", "cg1_comm_destroy",
    "cg1_validate_compressed", "cg1_fp_jacobi", "cg1_batch_decompress_pool", "cg1_batch_subgroup_pool", "cg1_batch_subgroup", "cg1_lincomb_batch", "cg1_lincomb_batch_pool", "cg1_glv_split", "cg1_batch_decompress_rows",
    "cg1_probe_add_chain", "cg1_msm_blobs", "cg1_stage_reserve", "cg1_msm_blobs_device", "cg1_vec_create", "cg1_vec_destroy", "cg1_vec_len", "cg1_msm_vec", "cg1_batch_normalize", "cg1_batch_from_affine96", "cg1_get_last_launches", "cg1_plan_describe",
]


class PinnedBuffer:
    """Page-locked host memory owned by a Context; `.buf` is a ctypes char array over it (pass it to native calls)."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        self.ptr = cg1_host_alloc(ctx.handle, max(1, self.nbytes))
        if not self.ptr:
            raise NativeError(f"hipHostMalloc of {nbytes} bytes failed")
        self.buf = (ctypes.c_char * max(1, self.nbytes)).from_address(self.ptr)

    def free(self) -> None:
        if self.ptr and self.ctx.handle:
            self.buf = None
            cg1_host_free(self.ctx.handle, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class DeviceBuffer:
    """A hipMalloc'd region owned by a Context (plain device pointer + size)."""

    def __init__(self, ctx: "Context", nbytes: int):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        self.ptr = cg1_dev_malloc(ctx.handle, self.nbytes)
        if not self.ptr:
            raise NativeError(f"hipMalloc of {nbytes} bytes failed")

    def upload(self, data: bytes, offset: int = 0) -> None:
        assert offset + len(data) <= self.nbytes
        self.ctx.check(cg1_h2d(self.ctx.handle, self.ptr + offset, data, len(data)))

    def download(self, nbytes: int | None = None, offset: int = 0) -> bytes:
        nbytes = self.nbytes - offset if nbytes is None else nbytes
        out = ctypes.create_string_buffer(nbytes)
        self.ctx.check(cg1_d2h(self.ctx.handle, out, self.ptr + offset, nbytes))
        return out.raw

    def free(self) -> None:
        if self.ptr:
            cg1_dev_free(self.ctx.handle, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context:
    """One per GPU: owns the HIP stream, the scratch buffers of the MSM pipeline and phase timers."""

    def __init__(self, device: int = 0, cu_mask=None):
        """cu_mask: iterable of CU indices the context's kernels may run on (None = the whole chip)."""
        self.device = device
        if cu_mask is None:
            self.handle = cg1_ctx_create(device)
        else:
            words = [0] * 16
            for cu in cu_mask:
                words[cu >> 5] |= 1 << (cu & 31)
            arr = (ctypes.c_uint32 * 16)(*words)
            self.handle = cg1_ctx_create_cu_mask(device, arr, 16)
        if not self.handle:
            raise NativeError(
                f"cg1_ctx_create({device}) failed: no MI355X/HIP device visible "
                f"(cg1_device_count() = {cg1_device_count()}).  The MSM path has no CPU fallback."
            )

    def check(self, rc: int) -> None:
        if rc != OK:
            msg = cg1_ctx_error(self.handle)
            raise NativeError(f"libcurdle_g1 error {rc}: {msg.decode() if msg else ''}")

    def alloc(self, nbytes: int) -> DeviceBuffer:
        return DeviceBuffer(self, nbytes)

    def sync(self) -> None:
        self.check(cg1_ctx_sync(self.handle))

    def set_param(self, name: str, value: int) -> None:
        self.check(cg1_ctx_set_param(self.handle, name.encode(), value))

    # ---- the hot path
    def msm_host(self, points_affine96: bytes, scalars32: bytes, n: int) -> bytes:
        assert len(points_affine96) >= 96 * n and len(scalars32) >= 32 * n
        out = ctypes.create_string_buffer(POINT_BYTES)
        self.check(cg1_msm(self.handle, points_affine96, scalars32, n, out))
        return out.raw

    def msm_affine(self, points_affine96_addr: int, scalars32_addr: int, n: int) -> bytes:
        """cg1_msm over raw host addresses (the Python face's page-locked staging)."""
        out = ctypes.create_string_buffer(POINT_BYTES)
        self.check(cg1_msm_addr(self.handle, points_affine96_addr, scalars32_addr, n, out))
        return out.raw

    def msm_blobs(self, blobs144, scalars32, n: int, all_normalised: bool = False) -> bytes:
        """compute_MSM over n host point blobs (bytes / ctypes buffer / raw address of page-locked memory) as G1Point objects hold them."""
        out = ctypes.create_string_buffer(POINT_BYTES)
        self.check(cg1_msm_blobs(self.handle, blobs144, scalars32, n, 1 if all_normalised else 0, out))
        return out.raw

    def stage_reserve(self, pts_bytes: int, sc_bytes: int):
        """(device address of the point staging, of the scalar staging) with room for that many bytes."""
        dp, ds = c_void_p(), c_void_p()
        self.check(cg1_stage_reserve(self.handle, pts_bytes, sc_bytes, ctypes.byref(dp), ctypes.byref(ds)))
        return dp.value, ds.value

    def h2d_async(self, dst_dev: int, src_host: int, nbytes: int) -> None:
        self.check(cg1_h2d_async(self.handle, dst_dev, src_host, nbytes))

    def copy_fence(self) -> None:
        self.check(cg1_copy_fence(self.handle))

    def msm_blobs_device(self, d_blobs144: int, d_scalars32: int, n: int, all_normalised: bool = False) -> bytes:
        out = ctypes.create_string_buffer(POINT_BYTES)
        self.check(cg1_msm_blobs_device(self.handle, d_blobs144, d_scalars32, n, 1 if all_normalised else 0, out))
        return out.raw

    def vec(self, blobs144, n: int, all_normalised: bool = False) -> "Vec":
        return Vec(self, blobs144, n, all_normalised)

    def msm_vec(self, vec: "Vec", scalars32, n: int, first: int = 0) -> bytes:
        out = ctypes.create_string_buffer(POINT_BYTES)
        self.check(cg1_msm_vec(self.handle, vec.handle, first, n, scalars32, out))
        return out.raw

    def msm_device(self, d_points, d_scalars, n: int, window_c: int = 0, shard_rank: int = 0, shard_world: int = 1) -> bytes:
        """d_points / d_scalars: DeviceBuffer or raw int device pointers (e.g. torch_tensor.data_ptr())."""
        p = d_points.ptr if isinstance(d_points, DeviceBuffer) else int(d_points)
        s = d_scalars.ptr if isinstance(d_scalars, DeviceBuffer) else int(d_scalars)
        out = ctypes.create_string_buffer(POINT_BYTES)
        self.check(cg1_msm_device(self.handle, p, s, n, window_c, shard_rank, shard_world, out))
        return out.raw

    def msm_device_begin(self, d_points, d_scalars, n: int, window_c: int = 0, shard_rank: int = 0, shard_world: int = 1) -> None:
        """Enqueue this context's whole MSM launch chain and return; msm_device_end() waits and delivers the point."""
        g = lambda b: b.ptr if isinstance(b, DeviceBuffer) else int(b)
        self.check(cg1_msm_device_begin(self.handle, g(d_points), g(d_scalars), n, window_c, shard_rank, shard_world))

    def msm_device_end(self) -> bytes:
        out = ctypes.create_string_buffer(POINT_BYTES)
        self.check(cg1_msm_device_end(self.handle, out))
        return out.raw

    def msm_batched_device(self, d_points, d_scalars, offsets, window_c: int = 0) -> list:
        """Regime B: offsets = [0, n_0, n_0+n_1, ...] (host ints); returns one 144-byte blob per MSM."""
        m = len(offsets) - 1
        p = d_points.ptr if isinstance(d_points, DeviceBuffer) else int(d_points)
        s = d_scalars.ptr if isinstance(d_scalars, DeviceBuffer) else int(d_scalars)
        arr = (ctypes.c_uint32 * (m + 1))(*offsets)
        out = ctypes.create_string_buffer(POINT_BYTES * max(m, 1))
        self.check(cg1_msm_batched_device(self.handle, p, s, arr, m, window_c, out))
        raw = out.raw
        return [raw[POINT_BYTES * j: POINT_BYTES * (j + 1)] for j in range(m)]

    def msm_batched_host(self, points_affine96: bytes, scalars32: bytes, offsets) -> list:
        m = len(offsets) - 1
        arr = (ctypes.c_uint32 * (m + 1))(*offsets)
        out = ctypes.create_string_buffer(POINT_BYTES * max(m, 1))
        self.check(cg1_msm_batched(self.handle, points_affine96, scalars32, arr, m, out))
        raw = out.raw
        return [raw[POINT_BYTES * j: POINT_BYTES * (j + 1)] for j in range(m)]

    def last_counts(self) -> dict:
        """Of the last MSM call: bucket entries (non-zero digits), chunks, and mixed additions = entries - chunks."""
        e, c = ctypes.c_uint32(), ctypes.c_uint32()
        cg1_get_last_counts(self.handle, ctypes.byref(e), ctypes.byref(c))
        return {"entries": int(e.value), "chunks": int(c.value), "mixed_adds": int(e.value) - int(c.value),
                "accumulate_launches": int(cg1_get_last_launches(self.handle))}

    def timer_begin(self) -> None:
        self.check(cg1_timer_begin(self.handle))

    def timer_end(self) -> float:
        ms = c_float()
        self.check(cg1_timer_end(self.handle, ctypes.byref(ms)))
        return float(ms.value)

    def timings(self) -> dict:
        ph = (c_float * NPHASE)()
        tail = c_float()
        c = c_int()
        cg1_get_timings(self.handle, ph, ctypes.byref(tail), ctypes.byref(c))
        d = {name: float(ph[i]) for i, name in enumerate(PHASE_NAMES)}
        d["host_tail"] = float(tail.value)
        d["window_c"] = int(c.value)
        hm = (c_float * 4)()
        cg1_get_host_timings(self.handle, hm)
        for i, name in enumerate(("host_enqueue", "host_wait", "host_events", "host_horner")):
            d[name] = float(hm[i])
        return d

    def batch_mul_device(self, d_bases, nbase: int, d_scalars, d_out, n: int) -> None:
        g = lambda b: b.ptr if isinstance(b, DeviceBuffer) else int(b)
        self.check(cg1_batch_mul_device(self.handle, g(d_bases), nbase, g(d_scalars), g(d_out), n))

    def batch_mul_add_host(self, bases96: bytes, nbase: int, scalars32: bytes, nscalars: int, addend96, n: int) -> bytes:
        """out[i] = addend[i] + scalars[i % nscalars] * bases[i % nbase] (host buffers; affine96 in/out)."""
        out = ctypes.create_string_buffer(96 * max(n, 1))
        self.check(cg1_batch_mul_add(self.handle, bases96, nbase, scalars32, nscalars, addend96, out, n))
        return out.raw[: 96 * n]

    def batch_decompress_host(self, data48: bytes, n: int, check_subgroup: bool = False) -> bytes:
        """n compressed48 -> n affine96 on the GPU; raises ValueError (like the wheel) on the first bad encoding."""
        out = ctypes.create_string_buffer(96 * max(n, 1))
        bad = c_size_t(0)
        rc = cg1_batch_decompress_gpu(self.handle, data48, out, n, 1 if check_subgroup else 0, ctypes.byref(bad))
        if rc in (ERR_ENCODING, ERR_NOT_ON_CURVE, ERR_NOT_IN_SUBGROUP):
            raise ValueError(f"Err From Rust: serialised data seems to be invalid (point {bad.value}, code {rc})")
        self.check(rc)
        return out.raw[: 96 * n]

    def gen_scalars_device(self, d_out, n: int, seed: int) -> None:
        p = d_out.ptr if isinstance(d_out, DeviceBuffer) else int(d_out)
        self.check(cg1_gen_scalars_device(self.handle, p, n, seed & 0xFFFFFFFFFFFFFFFF))

    def probe_mad_rate(self, waves_per_simd: int = 2, iters: int = 2000) -> float:
        """Chip-wide v_mad_u64_u32 rate in lane-operations per second (k_probe_mad_rate)."""
        v = ctypes.c_double()
        self.check(cg1_probe_mad_rate(self.handle, waves_per_simd, iters, ctypes.byref(v)))
        return float(v.value)

    def batch_sum_host(self, points_affine96: bytes, offsets) -> bytes:
        """Segmented point sum (crs.py:64-65): one affine96 record per group [offsets[j], offsets[j+1])."""
        m = len(offsets) - 1
        arr = (ctypes.c_uint32 * (m + 1))(*offsets)
        out = ctypes.create_string_buffer(96 * max(m, 1))
        self.check(cg1_batch_sum(self.handle, points_affine96, arr, m, out))
        return out.raw[: 96 * m]

    def probe_madd(self, d_points, npts: int, lanes: int, iters: int) -> float:
        p = d_points.ptr if isinstance(d_points, DeviceBuffer) else int(d_points)
        ms = c_float()
        self.check(cg1_probe_madd(self.handle, p, npts, lanes, iters, ctypes.byref(ms)))
        return float(ms.value)

    def close(self) -> None:
        if self.handle:
            cg1_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Vec:
    """A point vector resident on the device (cg1_vec): n prepared records made once from the host objects' blobs."""

    def __init__(self, ctx: Context, blobs144, n: int, all_normalised: bool = False):
        self.ctx, self.n = ctx, int(n)
        self.handle = cg1_vec_create(ctx.handle, blobs144, self.n, 1 if all_normalised else 0)
        if not self.handle:
            raise NativeError(f"cg1_vec_create({n} points) failed")

    def free(self) -> None:
        if self.handle:
            cg1_vec_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Staging:
    """Page-locked staging of the Python face on one context: the point blobs and scalars of a call are gathered out of the Python
    objects straight into it (csrc/pyface.c) and uploaded from it at full PCIe rate.  Grown geometrically, kept by the context."""

    def __init__(self, ctx: Context):
        self.ctx = ctx
        self.pts = None
        self.sc = None
        self.cap_pts = self.cap_sc = 0

    def points(self, n: int) -> int:
        if n > self.cap_pts:
            if self.pts is not None:
                self.pts.free()
            cap = max(1024, n + n // 4)
            self.pts = PinnedBuffer(self.ctx, POINT_BYTES * cap)
            self.cap_pts = cap
        return self.pts.ptr

    def scalars(self, n: int) -> int:
        if n > self.cap_sc:
            if self.sc is not None:
                self.sc.free()
            cap = max(1024, n + n // 4)
            self.sc = PinnedBuffer(self.ctx, 32 * cap)
            self.cap_sc = cap
        return self.sc.ptr


def msm_multi_device(ctxs, d_points, d_scalars, ns, window_c: int = 0) -> bytes:
    """One MSM over several GPUs of this process (cg1_msm_multi_device): ctxs[i] owns point shard i on its own device."""
    k = len(ctxs)
    assert k and len(d_points) == len(d_scalars) == len(ns) == k
    g = lambda b: b.ptr if isinstance(b, DeviceBuffer) else int(b)
    hs = (c_void_p * k)(*[c.handle for c in ctxs])
    ps = (c_void_p * k)(*[g(b) for b in d_points])
    ss = (c_void_p * k)(*[g(b) for b in d_scalars])
    nn = (c_size_t * k)(*[int(n) for n in ns])
    out = ctypes.create_string_buffer(POINT_BYTES)
    rc = cg1_msm_multi_device(hs, k, ps, ss, nn, window_c, out)
    if rc != OK:
        msgs = "; ".join((cg1_ctx_error(c.handle) or b"").decode() for c in ctxs)
        raise NativeError(f"libcurdle_g1 error {rc}: {msgs}")
    return out.raw


class Comm:
    """One rank's end of the N>1 exchange (cg1_comm_*): a TCP control channel on the loopback interface, plus RCCL once
    `attach_rccl(ctx)` has run on every rank.  See curdleproofs_pie_amd/distributed.py for the rendezvous."""

    def __init__(self, rank: int, world: int):
        self.rank, self.world = int(rank), int(world)
        self.handle = cg1_comm_create(self.rank, self.world)
        if not self.handle:
            raise NativeError(f"cg1_comm_create({rank}, {world}) failed")

    def check(self, rc: int) -> None:
        if rc != OK:
            msg = cg1_comm_error(self.handle)
            raise NativeError(f"libcurdle_g1 comm error {rc}: {msg.decode() if msg else ''}")

    @property
    def port(self) -> int:
        return int(cg1_comm_port(self.handle))

    @property
    def transport(self) -> str:
        return cg1_comm_transport(self.handle).decode()

    @property
    def world_seen(self) -> int:
        return int(cg1_comm_world_seen(self.handle))

    def connect(self, host: str, port: int, nonce: int, timeout_ms: int) -> int:
        return int(cg1_comm_connect(self.handle, host.encode() if host else None, int(port), int(nonce) & (2 ** 64 - 1), int(timeout_ms)))

    def set_timeout(self, timeout_ms: int) -> None:
        self.check(cg1_comm_set_timeout(self.handle, int(timeout_ms)))

    def attach_rccl(self, ctx: "Context") -> None:
        self._ctx = ctx          # the communicator keeps using ctx's stream: ctx must outlive it
        self.check(cg1_comm_attach_rccl(self.handle, ctx.handle))

    def allgather(self, data: bytes, host_only: bool = False) -> list:
        """Every rank passes the same number of bytes; returns the `world` byte strings in rank order."""
        n = len(data)
        out = ctypes.create_string_buffer(max(1, n * self.world))
        fn = cg1_comm_allgather_host if host_only else cg1_comm_allgather
        self.check(fn(self.handle, bytes(data), n, out))
        raw = out.raw
        return [raw[i * n:(i + 1) * n] for i in range(self.world)]

    def barrier(self) -> None:
        self.check(cg1_comm_barrier(self.handle))

    def allreduce_g1(self, partial_blob: bytes) -> bytes:
        out = ctypes.create_string_buffer(POINT_BYTES)
        self.check(cg1_comm_allreduce_g1(self.handle, bytes(partial_blob), out, None))
        return out.raw

    def close(self) -> None:
        if self.handle:
            cg1_comm_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None
_default_lock = threading.Lock()


_close_hooks: list = []


def on_close_default_context(fn) -> None:
    """Register `fn()` to run before the default context is closed (modules holding device / page-locked memory on it free it there)."""
    if fn not in _close_hooks:
        _close_hooks.append(fn)


def close_default_context() -> None:
    """Release the process-wide context (its stream, scratch and staging buffers); the next default_context() makes a new one."""
    global _default_ctx
    if _default_ctx is not None:
        for fn in list(_close_hooks):
            fn()
    with _default_lock:
        if _default_ctx is not None:
            _default_ctx.close()
            _default_ctx = None


def default_context() -> Context:
    """Process-wide context on the GPU named by CURDLE_G1_DEVICE / LOCAL_RANK (default 0)."""
    global _default_ctx
    with _default_lock:
        if _default_ctx is None:
            dev = int(os.environ.get("CURDLE_G1_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            count = cg1_device_count()
            if count > 0 and dev >= count and os.environ.get("CURDLE_G1_SHARE_DEVICE", "0") != "1":
                # a rank that silently lands on GPU 0 turns a scaling run into N processes on one card: say so instead
                # (CURDLE_G1_SHARE_DEVICE=1: rehearsals of several ranks on one GPU map rank r to device r mod count)
                raise NativeError(f"CURDLE_G1_DEVICE / LOCAL_RANK = {dev} but only {count} GPU(s) are visible "
                                  "(set CURDLE_G1_SHARE_DEVICE=1 to let ranks share devices)")
            if count > 0:
                dev %= count
            _default_ctx = Context(dev)
        return _default_ctx
