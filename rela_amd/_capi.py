"""ctypes binding of include/rela_amd.h (librela_amd.so).

There is no fallback: if the shared library is missing the import fails loudly.  Build it with
`python -m rela_amd.build` (or __graft_entry__.build()).
"""
import ctypes as C
import os

# PyTorch-ROCm bundles its own libamdhip64.so.7; librela_amd.so links the same SONAME.  Load
# torch FIRST so the process has exactly one HIP runtime (two copies = "0 devices visible").
import torch  # noqa: F401,E402

PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(PKG, "librela_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "rela_amd: %s is missing -- the HIP extension has not been built (python -m rela_amd.build). "
        "There is no CPU fallback for this path." % LIB_PATH)

lib = C.CDLL(LIB_PATH)

OK, EINVAL, ENODEV, ENOMEM, ESTATE, ESCAN, EWOULDBLOCK = 0, -1, -2, -3, -4, -5, -6

vp, i32, i64, u64, f32, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_float, C.c_double
P = C.POINTER


class ReplayState(C.Structure):
    _fields_ = [("head", C.c_int32), ("tail", C.c_int32), ("size", C.c_int32), ("safe_size", C.c_int32),
                ("ring", C.c_int32), ("n_sampled", C.c_int32), ("num_add", C.c_int64), ("sum", C.c_double),
                ("dev_error", C.c_int32), ("pad", C.c_int32)]


class ReplayIpcDesc(C.Structure):  # rela_replay_ipc_desc
    _fields_ = [("abi", C.c_int32), ("nfields", C.c_int32), ("ring", C.c_int32), ("device", C.c_int32),
                ("max_batch", C.c_int32), ("pad", C.c_int32), ("row_bytes", C.c_int64 * 16), ("steps", C.c_int32 * 16),
                ("field_handle", (C.c_ubyte * 64) * 16), ("ids_handle", C.c_ubyte * 64), ("raw_w_handle", C.c_ubyte * 64),
                ("state_handle", C.c_ubyte * 64)]


class ReplayChunkDesc(C.Structure):  # rela_replay_chunk_desc
    _fields_ = [("ipc", ReplayIpcDesc), ("abi", C.c_int32), ("nfds", C.c_int32), ("field_chunks", C.c_int32 * 16),
                ("chunk_bytes", C.c_int64 * 16), ("mapped_bytes", C.c_int64 * 16), ("dd_ups", C.c_int32),
                ("dd_field", C.c_int32 * 2), ("units_chunks", C.c_int32), ("dd_unit_bytes", C.c_int64), ("dd_cap", C.c_int64),
                ("units_chunk_bytes", C.c_int64), ("units_mapped_bytes", C.c_int64), ("units_handle", C.c_ubyte * 64)]


IPC_MAX_FDS = 128


class IpcAllreduceDesc(C.Structure):  # rela_ipc_allreduce_desc
    _fields_ = [("abi", C.c_int32), ("rank", C.c_int32), ("world", C.c_int32), ("device", C.c_int32),
                ("device_flags", C.c_int32), ("pad", C.c_int32), ("count", C.c_int64), ("bucket_offset", C.c_int64),
                ("bucket_handle", C.c_ubyte * 64), ("red_handle", C.c_ubyte * 64), ("shm_name", C.c_char * 64)]


class LSTMNetParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "conv3_w", "conv3_b", "w_ih",
                                          "w_hh", "b_ih", "b_hh", "v_w", "v_b", "a_w", "a_b")]


class FFNetParams(C.Structure):
    _fields_ = [(n, vp) for n in ("conv1_w", "conv1_b", "conv2_w", "conv2_b", "conv3_w", "conv3_b", "fc_w", "fc_b",
                                  "v_w", "v_b", "a_w", "a_b")]


MISSING = []  # symbols of include/rela_amd.h the loaded library lacks (tests assert this is empty)


def _sig(name, restype, argtypes):
    try:
        fn = getattr(lib, name)
    except AttributeError:
        MISSING.append(name)
        return None
    fn.restype = restype
    fn.argtypes = argtypes
    return fn


_sig("rela_last_error", C.c_char_p, [])
_sig("rela_abi_version", i32, [])
_sig("rela_stream_create", i32, [P(vp), i32])
_sig("rela_stream_destroy", None, [vp, i32])
_sig("rela_stream_synchronize", i32, [vp, i32])
_sig("rela_stream_wait_stream", i32, [vp, vp, i32])
_sig("rela_memcpy_h2d_async", i32, [vp, vp, i64, vp, i32])
_sig("rela_replay_set_decoupled_insert", i32, [vp, i32])
_sig("rela_replay_export_ipc", i32, [vp, vp])
_sig("rela_replay_import_ipc", i32, [P(vp), vp, i32])
_sig("rela_ipc_allreduce_create", i32, [P(vp), i32, i32, vp, i64, i32, i32, vp])
_sig("rela_ipc_allreduce_connect", i32, [vp, vp])
_sig("rela_ipc_allreduce_run", i32, [vp, vp])
_sig("rela_ipc_allreduce_mode", i32, [vp])
_sig("rela_ipc_allreduce_destroy", None, [vp])
_sig("rela_ipc_page_create", i32, [P(vp), C.c_char_p, i32])
_sig("rela_ipc_page_open", i32, [P(vp), C.c_char_p, i32])
_sig("rela_ipc_page_unlink", i32, [vp])
_sig("rela_ipc_page_close", None, [vp])
_sig("rela_ipc_page_host_ptr", vp, [vp])
_sig("rela_ipc_page_dev_ptr", vp, [vp])
_sig("rela_ipc_page_write32", i32, [vp, i32, C.c_uint32, vp])
_sig("rela_ipc_page_wait32", i32, [vp, i32, C.c_uint32, vp])
_sig("rela_ipc_page_host_store32", i32, [vp, i32, C.c_uint32])
_sig("rela_ipc_page_host_load32", i32, [vp, i32, P(C.c_uint32)])
_sig("rela_ipc_page_host_wait32", i32, [vp, i32, C.c_uint32, f64])
_sig("rela_ipc_page_selftest", i32, [vp, i32, C.c_uint32, P(i32)])
_sig("rela_ipc_alloc_buffer", i32, [P(vp), i64, i32])
_sig("rela_ipc_free_buffer", i32, [vp, i32])
_sig("rela_replay_set_chunk_bytes", i32, [vp, i64])
_sig("rela_runtime_set_replay_chunk_bytes", i32, [i64])
_sig("rela_replay_export_chunks", i32, [vp, vp, P(i32), i32])
_sig("rela_replay_import_chunks", i32, [P(vp), vp, P(i32), i32, i32])
_sig("rela_replay_remote_close", None, [vp])
_sig("rela_replay_remote_gather", i32, [vp, i32, vp, vp, vp, i32, i32, vp])
_sig("rela_ipc_export_buffer", i32, [vp, vp])
_sig("rela_ipc_import_buffer", i32, [vp, P(vp), i32])
_sig("rela_ipc_close_buffer", i32, [vp, i32])
_sig("rela_replay_create", i32, [P(vp), i32, i32, f32, f32, i32, i32])
_sig("rela_replay_destroy", None, [vp])
_sig("rela_replay_set_schema", i32, [vp, i32, P(i64)])
_sig("rela_replay_set_schema_seq", i32, [vp, i32, P(i64), P(C.c_int32)])
_sig("rela_replay_set_schema_dedup", i32, [vp, i32, P(i64), i32, i32, i64, i32, i64])
_sig("rela_replay_units_reserve", i32, [vp, i32, i32, P(i64), P(C.c_int32)])
_sig("rela_replay_units_write", i32, [vp, i64, i32, vp, i64, vp])
_sig("rela_replay_set_block_min_unit", i32, [vp, i32, i32, i64])
_sig("rela_replay_dedup_info", i32, [vp, P(i32), P(i64), P(i64)])
_sig("rela_replay_begin_add", i32, [vp, i32, i32, P(i32)])
_sig("rela_replay_write_rows", i32, [vp, i32, i32, i32, P(vp), vp])
_sig("rela_replay_write_rows_gather", i32, [vp, i32, i32, vp, P(vp), P(vp), vp])
_sig("rela_replay_commit_add", i32, [vp, i32, i32, vp, vp])
_sig("rela_replay_commit_add_grouped", i32, [vp, i32, i32, i32, vp, vp])
_sig("rela_replay_abort_add", i32, [vp, i32, i32])
_sig("rela_replay_add", i32, [vp, i32, P(vp), vp, i32, vp])
_sig("rela_replay_sample", i32, [vp, i32, P(vp), vp, vp])
_sig("rela_replay_update_priority", i32, [vp, i32, vp, i32, vp])
_sig("rela_replay_set_deferred_wait", i32, [vp, i32])
_sig("rela_replay_wait", i32, [vp, vp])
_sig("rela_replay_last_sample_dev", i32, [vp, P(vp), P(vp)])
_sig("rela_replay_last_sample_size", i32, [vp])
_sig("rela_replay_shutdown", i32, [vp])
_sig("rela_replay_limits", i32, [vp, P(i32), P(i32)])
_sig("rela_replay_size", i32, [vp])
_sig("rela_replay_num_add", i64, [vp])
_sig("rela_replay_debug_state", i32, [vp, P(ReplayState), vp, vp, vp])
_sig("rela_replay_debug_weights", i32, [vp, vp, vp])
_sig("rela_replay_debug_read_rows", i32, [vp, i32, i32, i32, vp])
_sig("rela_seqscan_search", i32, [vp, i64, i64, i64, vp, i32, vp, vp, vp, P(f64), vp])
_sig("rela_debug_pow", i32, [vp, i32, f32, vp, vp])
_sig("rela_seqscan_debug_perturb", i32, [i32])
_sig("rela_nstep_return", i32, [i32, i32, f32, i32, vp, vp, vp, vp, vp, vp])
_sig("rela_ffnet_create", i32, [P(vp), i32, i32])
_sig("rela_ffnet_destroy", None, [vp])
_sig("rela_ffnet_load", i32, [vp, P(FFNetParams), i32, vp])
_sig("rela_ffnet_num_action", i32, [vp])
_sig("rela_ffnet_version", C.c_uint64, [vp])
_sig("rela_ffnet_set_precision", i32, [vp, i32])
_sig("rela_ffnet_precision", i32, [vp])
_sig("rela_ffnet_debug_conv12_stamps", i32, [vp, i32, vp, vp, vp])
_sig("rela_ffnet_debug_fc_stamps", i32, [vp, i32, vp, vp, vp])
_sig("rela_ffnet_debug_conv3_stamps", i32, [vp, i32, vp, vp, vp])
_sig("rela_ffnet_debug_conv12_records", i32, [vp, i32, vp, vp, vp, vp, vp, vp])
_sig("rela_lstmnet_set_precision", i32, [vp, i32])
_sig("rela_lstmnet_precision", i32, [vp])
_sig("rela_ffnet_workspace_bytes", i64, [vp, i32])
_sig("rela_ffnet_forward", i32, [vp, i32, vp, vp, vp, vp, i64, vp])
_sig("rela_lstmnet_create", i32, [P(vp), i32, i32])
_sig("rela_lstmnet_destroy", None, [vp])
_sig("rela_lstmnet_load", i32, [vp, P(LSTMNetParams), i32, vp])
_sig("rela_lstmnet_num_action", i32, [vp])
_sig("rela_lstmnet_version", C.c_uint64, [vp])
_sig("rela_lstmnet_workspace_bytes", i64, [vp, i32])
_sig("rela_lstmnet_step", i32, [vp, i32, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, vp])
_sig("rela_apex_act_from_q", i32, [i32, i32, i32, vp, vp, vp, u64, u64, vp, vp])
_sig("rela_apex_td_from_q", i32, [i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, f32, vp, vp, vp])
_sig("rela_apex_actor_create", i32, [P(vp), i32, i32, i32, i32, f32, vp, u64, i32])
_sig("rela_apex_actor_destroy", None, [vp])
_sig("rela_apex_actor_obs_slot", vp, [vp])
_sig("rela_apex_actor_plane_stage", vp, [vp])
_sig("rela_apex_actor_slide_stacks", i32, [vp, vp, vp])
_sig("rela_apex_actor_eps_dev", vp, [vp])
_sig("rela_apex_actor_legal_dev", vp, [vp])
_sig("rela_apex_actor_act", i32, [vp, vp, vp, vp, vp, vp, P(vp), vp])
_sig("rela_apex_actor_post_step", i32, [vp, vp, vp, i32, vp, vp, i32, P(i32), vp])
_sig("rela_apex_actor_set_dedup", i32, [vp, i32])
_sig("rela_apex_actor_num_act", i64, [vp])
_sig("rela_apex_actor_set_reuse", i32, [vp, i32])
_sig("rela_apex_actor_last_q_dev", vp, [vp])
_sig("rela_apex_actor_last_priority_dev", vp, [vp])
_sig("rela_r2d2_actor_create", i32, [P(vp), i32, i32, i32, i32, f32, i32, i32, f64, vp, u64, i32])
_sig("rela_r2d2_actor_destroy", None, [vp])
_sig("rela_r2d2_actor_obs_slot", vp, [vp])
_sig("rela_r2d2_actor_plane_stage", vp, [vp])
_sig("rela_r2d2_actor_slide_stacks", i32, [vp, vp, vp])
_sig("rela_r2d2_actor_act", i32, [vp, vp, vp, vp, vp, vp, P(vp), vp])
_sig("rela_r2d2_actor_post_step", i32, [vp, vp, vp, vp, vp, i32, P(i32), vp])
_sig("rela_r2d2_actor_num_act", i64, [vp])
_sig("rela_r2d2_actor_set_reuse", i32, [vp, i32])
_sig("rela_r2d2_actor_hidden_dev", vp, [vp, i32])
_sig("rela_r2d2_actor_last_priority_dev", vp, [vp])
_sig("rela_apex_learner_create", i32, [P(vp), i32, i32, i32, f32, i32, f32, f32, f32, i32])
_sig("rela_apex_learner_destroy", None, [vp])
_sig("rela_apex_learner_load", i32, [vp, P(FFNetParams), P(FFNetParams), i32, vp])
_sig("rela_apex_learner_sync_target", i32, [vp, vp])
_sig("rela_apex_learner_set_precision", i32, [vp, i32])
_sig("rela_apex_learner_backward", i32, [vp, i32, P(vp), vp, vp, vp, vp])
_sig("rela_apex_learner_loss", i32, [vp, i32, P(vp), vp, vp, vp, vp])
_sig("rela_apex_learner_grad", i32, [vp, vp])
_sig("rela_apex_learner_apply", i32, [vp, vp])
_sig("rela_apex_learner_params", i32, [vp, P(FFNetParams), P(FFNetParams)])
_sig("rela_apex_learner_grads", i32, [vp, P(FFNetParams)])
_sig("rela_apex_learner_flat", i32, [vp, P(vp), P(vp), P(i64)])
_sig("rela_apex_learner_stats_dev", vp, [vp])
_sig("rela_apex_learner_debug_activations", i32, [vp, P(vp), P(vp), P(vp), P(vp), P(i32)])
_sig("rela_r2d2_learner_create", i32, [P(vp), i32, i32, i32, f32, i32, i32, f64, i32, f32, f32, f32, i32])
_sig("rela_r2d2_learner_destroy", None, [vp])
_sig("rela_r2d2_learner_load", i32, [vp, P(LSTMNetParams), P(LSTMNetParams), i32, vp])
_sig("rela_r2d2_learner_sync_target", i32, [vp, vp])
_sig("rela_r2d2_learner_backward", i32, [vp, i32, P(vp), vp, vp, vp, vp, vp])
_sig("rela_r2d2_learner_loss", i32, [vp, i32, P(vp), vp, vp, vp, vp, vp])
_sig("rela_r2d2_learner_grad", i32, [vp, vp])
_sig("rela_r2d2_learner_apply", i32, [vp, vp])
_sig("rela_r2d2_learner_params", i32, [vp, P(LSTMNetParams), P(LSTMNetParams)])
_sig("rela_r2d2_learner_grads", i32, [vp, P(LSTMNetParams)])
_sig("rela_r2d2_learner_flat", i32, [vp, P(vp), P(vp), P(i64)])
_sig("rela_r2d2_learner_stats_dev", vp, [vp])
_sig("rela_r2d2_learner_check", i32, [vp, vp])
_sig("rela_r2d2_learner_set_precision", i32, [vp, i32])
_sig("rela_prof_enable", i32, [i32])
_sig("rela_prof_set_filter", i32, [C.c_char_p])
_sig("rela_prof_summary_json", i32, [C.c_char_p, i64])
_sig("rela_prof_count_enable", i32, [i32])
_sig("rela_runtime_set_cu_reserve", i32, [i32])
_sig("rela_prof_counts_json", i32, [C.c_char_p, i64])


class RelaError(RuntimeError):
    def __init__(self, code, where):
        msg = lib.rela_last_error()
        super().__init__("%s failed with code %d: %s" % (where, code, msg.decode() if msg else ""))
        self.code = code


def check(code, where):
    if code != OK:
        raise RelaError(code, where)


class launch_census:
    """`with launch_census() as c: ...; c.counts` -- the kernels that REALLY ran inside the block, by name
    (rela_prof_count_enable / rela_prof_counts_json): the parity tests of a fast mode assert on it."""

    def __enter__(self):
        check(lib.rela_prof_count_enable(1), "rela_prof_count_enable")
        self.counts = {}
        return self

    def __exit__(self, *exc):
        import json

        buf = C.create_string_buffer(1 << 16)
        check(lib.rela_prof_counts_json(buf, len(buf)), "rela_prof_counts_json")
        lib.rela_prof_count_enable(0)
        self.counts = json.loads(buf.value.decode())
        return False
