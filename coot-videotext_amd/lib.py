"""
ctypes binding of libcoot_hip.so (C ABI declared in include/coot_hip.h).

The product path has NO CPU fallback: if the shared library is missing or a call fails, a
RuntimeError is raised (same convention as the reference, which raises plain Python exceptions,
e.g. nntrainer/models/transformer_legacy.py:154-156, :252).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Tuple

_HERE = os.path.dirname(os.path.abspath(__file__))
# Two builds of the same sources ship (csrc/build.sh, csrc/common.h): libcoot_hip.so computes on bfloat16 MFMA operands (the default,
# what bench.py times), libcoot_hip_f16.so on IEEE half operands — the reference's GPU arithmetic (fp16 autocast,
# coot/trainer_retrieval.py:264; BASELINE.json configs[3]) — forward-only.  COOT_OPERAND=f16 selects the second one for the process
# (one library per process); COOT_HIP_LIB: any other build (tools/build_variant.sh: A/B of compile-time kernel variants on one GPU box).
OPERAND_ENV = os.environ.get("COOT_OPERAND", "bf16")
if OPERAND_ENV not in ("bf16", "f16"):
    raise RuntimeError(f"COOT_OPERAND={OPERAND_ENV}: bf16 or f16")
LIB_PATH = os.environ.get("COOT_HIP_LIB") or os.path.join(_HERE, "lib", "libcoot_hip_f16.so" if OPERAND_ENV == "f16" else "libcoot_hip.so")

EXPORTS = [
    "coot_last_error", "coot_version", "coot_set_option", "coot_get_option", "coot_debug_timestamps", "coot_debug_step_stamps", "coot_debug_dropout_scales", "coot_debug_attn_dropout_scales", "coot_net_param_numel", "coot_net_param_count",
    "coot_net_param_info", "coot_net_out_dim", "coot_net_wpack_bytes", "coot_net_pack_weights", "coot_nets_pack_weights",
    "coot_net_saved_bytes", "coot_net_scratch_bytes", "coot_net_fwd", "coot_net_bwd", "coot_net_grads_overwrite", "coot_nets_zero_grads", "coot_nets_zero_grads_ex", "coot_debug_written_matrices", "coot_pack_fwd",
    "coot_pack_bwd", "coot_contrastive_scratch_bytes", "coot_contrastive_fwd_bwd", "coot_contrastive_fwd_bwd_part", "coot_contrastive_f32_scratch_bytes", "coot_contrastive_fwd_bwd_f32", "coot_cyclecons_fwd_bwd",
    "coot_gemm_nt", "coot_gemm_tn", "coot_gemm_tn_batch", "coot_debug_tn_xcd_map", "coot_debug_clock_monitor", "coot_gemm_tn_workspace_bytes", "coot_ln_fwd", "coot_attn_fwd", "coot_probe_tr16", "coot_timing_enable",
    "coot_timing_collect", "coot_step_workspace_bytes", "coot_train_step", "coot_step_forward", "coot_step_backward",
    "coot_adam_step", "coot_radam_step", "coot_step_update", "coot_step_set_global_done_events", "coot_step_device_state_bytes", "coot_step_set_device_state", "coot_collate_level", "coot_collate_packed", "coot_sample_cycle_indices", "coot_step_set_cycle_indices", "coot_step_input_stage_bytes", "coot_step_set_input_stages", "coot_step_set_next_batch", "coot_contrastive_fwd_bwd_dp", "coot_contrastive_fwd_bwd_dp_blocks", "coot_retrieval_workspace_bytes", "coot_retrieval_ranks",
    "coot_det_shadow_bytes", "coot_det_configure", "coot_det_flush", "coot_event_record", "coot_event_wait", "coot_event_handle", "coot_stream_hop",
    "coot_stream_create_concurrent", "coot_stream_destroy", "coot_streams_overlap",
]


class NetConfig(C.Structure):
    """coot_net_config (include/coot_hip.h)."""
    _fields_ = [("input_dim", C.c_int), ("hidden_dim", C.c_int), ("num_heads", C.c_int), ("ff_dim", C.c_int),
                ("num_layers", C.c_int), ("use_input_fc", C.c_int), ("use_context", C.c_int),
                ("ctx_num_layers", C.c_int), ("pooler", C.c_int), ("pool_hidden", C.c_int), ("pool_heads", C.c_int),
                ("dropout", C.c_float), ("ctx_dropout", C.c_float), ("pool_dropout", C.c_float), ("dtype", C.c_int)]


class ContrastiveConfig(C.Structure):
    """coot_contrastive_config."""
    _fields_ = [("margin", C.c_float), ("weight_high", C.c_float), ("weight_high_internal", C.c_float),
                ("weight_low", C.c_float), ("weight_low_internal", C.c_float), ("weight_context", C.c_float),
                ("weight_context_internal", C.c_float)]


SOURCE_PADDED, SOURCE_PACKED_F32, SOURCE_PACKED_BF16 = 0, 1, 2  # COOT_SOURCE_* (include/coot_hip.h)
DTYPE_BF16, DTYPE_F32, DTYPE_F16 = 0, 1, 2  # COOT_DTYPE_* (coot_net_config.dtype)
STEP_OPTIMIZER, STEP_REPACK, STEP_PACKS_FRESH, STEP_DEFER_TEXT_JOIN, STEP_INPUT_STAGES, STEP_STAGE_ANNOUNCED = 1, 2, 4, 8, 16, 32  # coot_train_step do_optimizer bits (include/coot_hip.h)
FWD_PACKS_FRESH, FWD_INPUT_STAGES, FWD_STAGE_ANNOUNCED = 1, 2, 4  # coot_step_forward packs_fresh bits
UPDATE_REPACK, UPDATE_DEFER_TEXT_JOIN, UPDATE_SKIP_GLOBAL, UPDATE_GLOBAL_ONLY = 1, 2, 4, 8  # coot_step_update repack bits
DP_MAX_RANKS = 16  # COOT_DP_MAX_RANKS: ranks whose gathered blocks coot_contrastive_fwd_bwd_dp_blocks addresses in place


class StepConfig(C.Structure):
    """coot_step_config."""
    _fields_ = [("net", NetConfig * 4), ("contr", ContrastiveConfig), ("cc_weight", C.c_float), ("lr", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float), ("weight_decay", C.c_float),
                ("optimizer", C.c_int), ("radam_degentosgd", C.c_int)]


class StepDims(C.Structure):
    """coot_step_dims; tok_vis / tok_txt: packed token totals (0 = padded layout); source: SOURCE_* (who holds the token rows)."""
    _fields_ = [(n, C.c_int) for n in ("B", "Nc", "Lv", "Lc", "Lp", "Ls", "Cmax_clip", "Cmax_sent", "tok_vis", "tok_txt", "source")]


class TnProblem(C.Structure):
    """coot_tn_problem: one weight-gradient GEMM C[Mo, No] (+)= A[T, Mo]^T . B[T, No] of a batched launch."""
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64), ("T", C.c_int), ("Mo", C.c_int),
                ("No", C.c_int), ("C", C.c_void_p), ("ldc", C.c_int64), ("a_colsum", C.c_void_p), ("overwrite", C.c_int),
                ("groups", C.c_int), ("zA", C.c_int64), ("zB", C.c_int64), ("zC", C.c_int64)]


class PackedSeqs(C.Structure):
    """coot_packed_seqs: device int32 row starts [nseq + 1] + their last entry on the host."""
    _fields_ = [("cu_seqlens", C.c_void_p), ("total_tokens", C.c_int), ("source", C.c_int)]


class StepBuffers(C.Structure):
    _fields_ = [(n, C.c_void_p * 4) for n in ("params", "grads", "wpack", "adam_m", "adam_v", "decay_mask", "pe", "decay_block_all")]


class StepBatch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("vid_feat", "clip_feat", "par_feat", "sent_feat", "vid_len", "clip_len",
                                          "par_len", "sent_len", "clip_num", "sent_num", "cu_vis", "cu_txt")]


_lib = None
ABI_VERSION = 7  # include/coot_hip.h: COOT_ABI_VERSION


class ConcurrentStream:
    """A HIP stream whose kernels really overlap those of `others` (coot_stream_create_concurrent: HIP maps a process's streams onto
    a few hardware queues in creation order, and two streams on one queue run one after the other — include/coot_hip.h), wrapped
    for torch (`with torch.cuda.stream(s.torch)`, s.cuda_stream).  priority > 0: the device's lowest.  close() destroys it."""

    def __init__(self, others, priority: int = 0):
        import torch
        lib = load()
        ptrs = [int(getattr(o, "cuda_stream", o) or 0) for o in others]
        arr = (C.c_void_p * max(len(ptrs), 1))(*ptrs)
        out, conc = C.c_void_p(), C.c_int(0)
        check(lib.coot_stream_create_concurrent(arr, len(ptrs), int(priority), C.byref(out), C.byref(conc)), "coot_stream_create_concurrent")
        self.cuda_stream = out.value
        self.concurrent = bool(conc.value)
        self.torch = torch.cuda.ExternalStream(self.cuda_stream)

    def close(self) -> None:
        if self.cuda_stream is not None and _lib is not None:
            _lib.coot_stream_destroy(self.cuda_stream)
        self.cuda_stream = None


def operand() -> str:
    """The 16-bit MFMA operand format of the LOADED library: "bf16" or "f16" (coot_get_option("operand_f16"))."""
    v = C.c_int(0)
    check(load().coot_get_option(b"operand_f16", C.byref(v)), "coot_get_option")
    return "f16" if v.value else "bf16"


def build_hint() -> str:
    return f"build it with: bash {os.path.join(_HERE, 'csrc', 'build.sh')}  (or python -c 'import __graft_entry__ as g; g.build()')"


def load():
    """Load libcoot_hip.so (once).  Raises RuntimeError if it is missing — there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"libcoot_hip.so not found at {LIB_PATH}; {build_hint()}")
    lib = C.CDLL(LIB_PATH)
    vp, i32, i64, f32, sz, u64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t, C.c_uint64
    cfgp = C.POINTER(NetConfig)
    lib.coot_last_error.restype = C.c_char_p
    lib.coot_last_error.argtypes = []
    lib.coot_version.restype = i32
    lib.coot_version.argtypes = []
    if lib.coot_version() != ABI_VERSION:  # struct layouts and flag words of include/coot_hip.h (COOT_ABI_VERSION) this binding was written for
        raise RuntimeError(f"{LIB_PATH} has ABI version {lib.coot_version()}, this binding needs {ABI_VERSION}; rebuild: {build_hint()}")
    lib.coot_set_option.argtypes = [C.c_char_p, i32]
    lib.coot_get_option.argtypes = [C.c_char_p, C.POINTER(i32)]
    lib.coot_debug_timestamps.argtypes = [vp]
    lib.coot_debug_step_stamps.argtypes = [C.c_char_p, i32]
    lib.coot_net_param_numel.restype = i64
    lib.coot_net_grads_overwrite.argtypes = [i32]
    lib.coot_debug_written_matrices.argtypes = [cfgp, C.POINTER(i64), C.POINTER(i64), i32]
    lib.coot_nets_zero_grads_ex.argtypes = [i32, C.POINTER(cfgp), C.POINTER(vp), i32, C.POINTER(vp), C.POINTER(i64), i32, vp]
    lib.coot_net_param_numel.argtypes = [cfgp]
    lib.coot_net_param_count.argtypes = [cfgp]
    lib.coot_net_param_info.argtypes = [cfgp, i32, C.c_char_p, i32, C.POINTER(i64), C.POINTER(i64), C.POINTER(i32)]
    lib.coot_net_out_dim.argtypes = [cfgp]
    lib.coot_net_wpack_bytes.restype = sz
    lib.coot_net_wpack_bytes.argtypes = [cfgp]
    lib.coot_net_pack_weights.argtypes = [cfgp, vp, vp, vp]
    lib.coot_nets_pack_weights.argtypes = [i32, vp, vp, vp, vp]
    lib.coot_net_saved_bytes.restype = sz
    lib.coot_net_saved_bytes.argtypes = [cfgp, i32, i32, i32, i32]
    lib.coot_net_scratch_bytes.restype = sz
    lib.coot_net_scratch_bytes.argtypes = [cfgp, i32, i32, i32, i32]
    pkp = C.POINTER(PackedSeqs)
    lib.coot_net_fwd.argtypes = [cfgp, vp, vp, vp, vp, vp, i32, i32, vp, vp, i32, i32, vp, vp, vp, vp, sz, vp, sz, i32, u64, vp, vp, pkp]
    lib.coot_net_bwd.argtypes = [cfgp, vp, vp, vp, vp, vp, i32, i32, vp, vp, i32, i32, vp, vp, vp, vp, vp, vp, sz, vp, sz, i32, u64, vp, vp, pkp]
    lib.coot_pack_fwd.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp]
    lib.coot_pack_bwd.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    lib.coot_contrastive_scratch_bytes.restype = sz
    lib.coot_contrastive_scratch_bytes.argtypes = [i32, i32, i32, i32]
    lib.coot_contrastive_fwd_bwd.argtypes = [C.POINTER(ContrastiveConfig), i32, i32, i32, i32] + [vp] * 6 + [vp] + [vp] * 6 + [vp, sz, vp]
    lib.coot_contrastive_fwd_bwd_part.argtypes = lib.coot_contrastive_fwd_bwd.argtypes[:-1] + [i32, vp]
    lib.coot_contrastive_f32_scratch_bytes.restype = sz
    lib.coot_contrastive_f32_scratch_bytes.argtypes = [i32, i32, i32, i32]
    lib.coot_contrastive_fwd_bwd_f32.argtypes = list(lib.coot_contrastive_fwd_bwd.argtypes)
    lib.coot_cyclecons_fwd_bwd.argtypes = [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, f32, vp, vp, vp, vp, vp, vp]
    lib.coot_retrieval_workspace_bytes.argtypes = [i32, i32]
    lib.coot_retrieval_workspace_bytes.restype = C.c_size_t
    lib.coot_retrieval_ranks.argtypes = [vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, C.c_size_t, vp]
    lib.coot_det_shadow_bytes.restype = sz
    lib.coot_det_shadow_bytes.argtypes = [i32, C.POINTER(sz)]
    lib.coot_det_configure.argtypes = [i32, C.POINTER(vp), C.POINTER(sz), vp, sz, vp]
    lib.coot_det_flush.argtypes = [vp, sz, vp]
    lib.coot_event_record.argtypes = [i32, vp]
    lib.coot_event_wait.argtypes = [i32, vp]
    lib.coot_event_handle.restype = vp
    lib.coot_event_handle.argtypes = [i32]
    lib.coot_stream_hop.argtypes = [vp, vp]
    lib.coot_stream_create_concurrent.argtypes = [C.POINTER(vp), i32, i32, C.POINTER(vp), C.POINTER(i32)]
    lib.coot_stream_destroy.argtypes = [vp]
    lib.coot_streams_overlap.argtypes = [vp, vp]
    lib.coot_gemm_nt.argtypes = [vp, i64, vp, i64, i32, i32, i32, vp, i32, vp, i64, vp, i64, i32, vp]
    lib.coot_gemm_tn.argtypes = [vp, i64, vp, i64, i32, i32, i32, vp, i64, vp, sz, vp]
    lib.coot_gemm_tn_batch.argtypes = [C.POINTER(TnProblem), i32, vp, sz, vp, vp]
    ip = C.POINTER(i32)
    lib.coot_debug_tn_xcd_map.argtypes = [i32, ip, ip, ip, ip, ip, ip, i32]
    lib.coot_debug_clock_monitor.argtypes = [vp, i32, i32, vp]
    lib.coot_gemm_tn_workspace_bytes.restype = sz
    lib.coot_gemm_tn_workspace_bytes.argtypes = [i32, i32, i32]
    lib.coot_ln_fwd.argtypes = [vp, i32, i32, vp, vp, vp, vp, vp]
    lib.coot_attn_fwd.argtypes = [vp, i32, i32, i32, i32, vp, vp, vp, vp]
    lib.coot_probe_tr16.argtypes = [vp, vp]
    lib.coot_timing_enable.argtypes = [i32]
    lib.coot_timing_collect.argtypes = [i32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(i32)]
    scp, sdp, sbp, sxp = C.POINTER(StepConfig), C.POINTER(StepDims), C.POINTER(StepBuffers), C.POINTER(StepBatch)
    lib.coot_step_workspace_bytes.restype = sz
    lib.coot_step_workspace_bytes.argtypes = [scp, sdp]
    lib.coot_train_step.argtypes = [scp, sbp, sxp, sdp, vp, vp, sz, i32, u64, i64, i32, vp, vp, vp]
    lib.coot_step_forward.argtypes = [scp, sbp, sxp, sdp] + [vp] * 6 + [vp, sz, i32, u64, i32, vp, vp, vp]
    lib.coot_step_update.argtypes = [scp, sbp, i64, i32, vp, vp, vp, vp]
    lib.coot_step_set_global_done_events.argtypes = [vp, vp]
    lib.coot_step_device_state_bytes.restype = sz
    lib.coot_step_device_state_bytes.argtypes = []
    lib.coot_step_set_device_state.argtypes = [vp]
    lib.coot_collate_level.argtypes = [vp, vp, i64, i64, i64, i32, vp, vp, i32]
    lib.coot_collate_packed.argtypes = [vp, vp, i64, i64, i32, vp, vp, i32]
    lib.coot_sample_cycle_indices.argtypes = [vp, vp, i32, u64, vp, vp]
    lib.coot_step_set_cycle_indices.argtypes = [vp]
    lib.coot_step_input_stage_bytes.argtypes = [scp, sdp]
    lib.coot_step_input_stage_bytes.restype = sz
    lib.coot_step_set_input_stages.argtypes = [vp, vp, sz]
    lib.coot_step_set_next_batch.argtypes = [sxp, sdp]
    lib.coot_contrastive_fwd_bwd_dp.argtypes = [C.POINTER(ContrastiveConfig), i32, i32, i32, i32, C.POINTER(vp * 6), C.POINTER(i64 * 6), vp,
                                                C.POINTER(vp * 6), i32, i32, i32, i32, vp, sz, vp]
    lib.coot_contrastive_fwd_bwd_dp_blocks.argtypes = [C.POINTER(ContrastiveConfig), i32, i32, vp, vp, i32, i32, vp, vp, C.POINTER(i64 * 6), vp,
                                                       C.POINTER(vp * 6), vp, sz, vp]
    lib.coot_step_backward.argtypes = [scp, sbp, sxp, sdp] + [vp] * 10 + [vp, sz, i32, u64, vp, vp, vp]
    lib.coot_adam_step.argtypes = [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i64, vp]
    lib.coot_radam_step.argtypes = [vp, vp, vp, vp, vp, i64, f32, f32, f32, f32, f32, i64, i32, vp]
    _lib = lib
    # A/B switches from the environment (COOT_OPTIONS="name=value,name=value": coot_set_option), e.g. to run the test-suite on an
    # alternative kernel path
    for kv in filter(None, os.environ.get("COOT_OPTIONS", "").split(",")):
        k, v = kv.split("=")
        if lib.coot_set_option(k.strip().encode(), int(v)) != 0:
            raise RuntimeError(f"COOT_OPTIONS: {lib.coot_last_error().decode()}")
    return lib


def get_option(name: str) -> int:
    v = C.c_int(0)
    check(load().coot_get_option(name.encode(), C.byref(v)), "get_option")
    return v.value


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().coot_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"libcoot_hip {what} failed (rc={rc}): {msg}")


def ptr(t) -> int:
    """Device pointer of a torch tensor (or None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream


def param_table(cfg: NetConfig) -> Tuple[int, List[Tuple[str, int, Tuple[int, ...]]]]:
    """(numel, [(state-dict name, offset, shape)]) of the flat fp32 parameter arena."""
    lib = load()
    total = lib.coot_net_param_numel(C.byref(cfg))
    if total < 0:
        check(-1, "coot_net_param_numel")
    n = lib.coot_net_param_count(C.byref(cfg))
    out = []
    name = C.create_string_buffer(256)
    off = C.c_int64()
    shape = (C.c_int64 * 4)()
    ndim = C.c_int()
    for i in range(n):
        check(lib.coot_net_param_info(C.byref(cfg), i, name, 256, C.byref(off), shape, C.byref(ndim)), "coot_net_param_info")
        out.append((name.value.decode(), int(off.value), tuple(int(shape[j]) for j in range(ndim.value))))
    return int(total), out
