"""ctypes binding of ``librlpyt_hip.so`` (C ABI declared in ``include/rlpyt_hip.h``).

This is the stub a maintainer of the reference would add: the
reference is pure Python (no FFI of its own), so each entry point here replaces the body
of one reference routine.  There is NO CPU fallback: if the shared library is missing the
import fails loudly, and every wrapper raises on CPU-only torch builds when asked to run.

``torch`` is imported first on purpose: it loads its bundled ``libamdhip64.so`` (soname
``libamdhip64.so.7``) so that the kernels here share torch's HIP runtime, streams and
device allocations instead of instantiating a second runtime.
"""
import ctypes
import os
from ctypes import (POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_uint32, c_void_p)

import torch  # noqa: F401  (must precede the CDLL load; see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# (RLPYT_HIP_LIB: another build of the same ABI, for A/B runs of two kernel versions on one box)
LIB_PATH = os.environ.get("RLPYT_HIP_LIB") or os.path.join(_HERE, "csrc", "librlpyt_hip.so")
ABI_VERSION = 17


class CopyDesc(ctypes.Structure):
    _fields_ = [("dst", c_void_p), ("src", c_void_p), ("nbytes", c_int64)]


class StepGroup(ctypes.Structure):
    """Mirror of ``rlpyt_step_group`` (include/rlpyt_hip.h)."""
    _fields_ = [("act_word", c_void_p), ("obs_word", c_void_p), ("acts", c_uint32),
                ("rounds", c_uint32), ("n_workers", ctypes.c_int32), ("n_h2d", ctypes.c_int32),
                ("h2d", CopyDesc * 8), ("n_d2h", ctypes.c_int32), ("dedup", ctypes.c_int32),
                ("d2h", CopyDesc * 4), ("Bg", ctypes.c_int32), ("reserved", ctypes.c_int32),
                ("reset_flags", c_void_p), ("slot_host", c_void_p), ("full_rows_dev", c_void_p),
                ("obs_host", c_void_p), ("row_bytes", c_int64), ("t_host", c_void_p),
                ("graph_exec", c_void_p),
                ("stream", c_void_p), ("event", c_void_p), ("tail_graph_exec", c_void_p)]


class AppendField(ctypes.Structure):
    """Mirror of ``rlpyt_append_field`` (include/rlpyt_hip.h)."""
    _fields_ = [("ring", c_void_p), ("src", c_void_p), ("row_bytes", c_int64)]


class AdamTensor(ctypes.Structure):
    """Mirror of ``rlpyt_adam_tensor`` (include/rlpyt_hip.h)."""
    _fields_ = [("p", c_void_p), ("g", c_void_p), ("m", c_void_p), ("v", c_void_p), ("n", c_int64)]


class HipExtensionMissing(ImportError):
    pass


class HipError(RuntimeError):
    pass


def _load():
    if not os.path.exists(LIB_PATH):
        raise HipExtensionMissing(
            f"{LIB_PATH} not found: the MI355X HIP extension is not built. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C "
            "rlpyt_amd/csrc`). There is deliberately no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    return lib


lib = _load()

_p = c_void_p
_SIGNATURES = {
    # name: (restype, [argtypes])
    "rlpyt_hip_last_error": (c_char_p, []),
    "rlpyt_hip_abi_version": (c_int, []),
    "rlpyt_hip_device_info": (c_int, [c_char_p, c_int]),
    "rlpyt_hip_last_variant": (c_char_p, []),
    "rlpyt_hip_variant_reset": (None, []),
    "rlpyt_hip_variant_dump": (c_int64, [c_char_p, c_int64]),
    "rlpyt_host_register": (c_int, [_p, c_int64]),
    "rlpyt_host_unregister": (c_int, [_p]),
    "rlpyt_host_device_pointer": (c_int, [_p, POINTER(c_void_p)]),
    "rlpyt_seq_wait": (c_int, [_p, c_uint32, c_int, c_int]),
    "rlpyt_seq_post": (c_int, [_p, c_uint32]),
    "rlpyt_seq_arrive": (c_int, [_p, c_uint32]),
    "rlpyt_gae_f32": (c_int, [_p, _p, _p, _p, _p, _p, _p, c_int, c_int64, c_double, c_double,
                              c_int, _p]),
    "rlpyt_discount_return_f32": (c_int, [_p, _p, _p, _p, _p, _p, _p, c_int, c_int64, c_double,
                                          c_int, _p]),
    "rlpyt_valid_from_done": (c_int, [_p, _p, c_int, c_int64, _p]),
    "rlpyt_nstep_return_f32": (c_int, [_p, _p, _p, _p, c_int, c_int64, c_int, c_double, c_int,
                                       _p]),
    "rlpyt_adv_normalize_workspace_bytes": (c_int64, [c_int64]),
    "rlpyt_adv_normalize_f32": (c_int, [_p, _p, c_int64, c_float, _p, _p, _p]),
    "rlpyt_pg_loss_workspace_bytes": (c_int64, [c_int64]),
    "rlpyt_ppo_loss_fwd_bwd_f32": (c_int, [_p, _p, _p, _p, _p, _p, _p, c_int64, c_int, c_float,
                                           c_float, c_float, _p, _p, _p, _p, _p]),
    "rlpyt_ppo_head_loss_workspace_bytes": (c_int64, [c_int, c_int]),
    "rlpyt_ppo_head_loss_fwd_bwd_f32": (c_int, [_p] * 11 + [c_int, c_int64, c_int64, c_int, c_int,
                                                          c_float, c_float, c_float, _p, _p, _p, _p,
                                                          _p]),
    "rlpyt_ppo_trunk_head_loss_fwd_bwd_f32": (c_int, [_p] * 12 + [c_int, c_int64, c_int64, c_int, c_int,
                                                          c_float, c_float, c_float, _p, _p, _p, _p,
                                                          _p]),
    "rlpyt_ppo_trunk_head_loss_fwd_bwd_dev_f32": (c_int, [_p] * 12 + [c_int, c_int64, c_int64, c_int,
                                                              c_int, c_float, _p, c_float, c_float,
                                                              _p, _p, _p, _p, _p]),
    "rlpyt_gemm_nt_f32": (c_int, [_p, _p, _p, c_int64, c_int64, c_int64, _p]),
    "rlpyt_gemm_tn_workspace_bytes": (c_int64, [c_int64, c_int64, c_int64]),
    "rlpyt_gemm_tn_f32": (c_int, [_p, _p, _p, c_int64, c_int64, c_int64, _p, _p]),
    "rlpyt_a2c_loss_fwd_bwd_f32": (c_int, [_p, _p, _p, _p, _p, _p, c_int64, c_int, c_float,
                                           c_float, _p, _p, _p, _p, _p]),
    "rlpyt_dqn_loss_fwd_bwd_f32": (c_int, [_p, _p, _p, _p, _p, _p, _p, c_int64, c_int, c_float,
                                           c_float, _p, _p, _p, _p, _p]),
    "rlpyt_r2d1_loss_workspace_bytes": (c_int64, []),
    "rlpyt_r2d1_loss_fwd_bwd_f32": (c_int, [_p, _p, _p, _p, _p, _p, _p, _p, c_int, c_int, c_int,
                                            c_float, c_float, c_float, c_float, _p, _p, _p, _p,
                                            _p, _p]),
    "rlpyt_cat_dqn_loss_workspace_bytes": (c_int64, []),
    "rlpyt_cat_dqn_loss_fwd_bwd_f32": (c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, c_int64, c_int,
                                               c_int, c_float, c_float, c_float, _p, _p, _p, _p,
                                               _p]),
    "rlpyt_obs_rms_workspace_bytes": (c_int64, [c_int64, c_int64]),
    "rlpyt_obs_batch_stats_f32": (c_int, [_p, c_int64, c_int64, _p, _p, _p, _p]),
    "rlpyt_obs_rms_merge_f32": (c_int, [_p, _p, _p, _p, _p, c_float, c_int64, _p]),
    "rlpyt_obs_normalize_f32": (c_int, [_p, _p, _p, _p, c_int64, c_int64, c_float, c_float,
                                        _p]),
    "rlpyt_gather_tb": (c_int, [_p, _p, _p, c_int, c_int64, c_int64, c_int64, _p]),
    "rlpyt_obs_to_nhwc_f32": (c_int, [_p, _p, _p, c_int, c_int64, c_int, c_int64, c_int64,
                                      c_float, _p]),
    "rlpyt_commit_rows": (c_int, [_p, c_int, c_int64, _p, _p]),
    "rlpyt_categorical_head_f32": (c_int, [_p, _p, _p, _p, _p, _p, _p, c_int64, c_int, c_int, _p,
                                           _p, _p, _p]),
    "rlpyt_sampler_serve": (c_int, [_p, c_int, c_int, c_int, c_int, c_int, _p]),
    "rlpyt_fc_small_workspace_bytes": (c_int64, [c_int, c_int]),
    "rlpyt_fc_small_f32": (c_int, [_p, _p, _p, _p, c_int, c_int, c_int, c_int, _p, _p]),
    "rlpyt_frame_push": (c_int, [_p, _p, c_int64, c_int64, c_int64, c_int, c_int64, _p, _p, _p, _p,
                                 _p, _p, _p, _p, _p]),
    "rlpyt_fc_small_ksplit": (c_int, [c_int]),
    "rlpyt_eps_greedy_f32": (c_int, [_p, c_int64, c_int, _p, c_int, _p, _p, _p, _p]),
    "rlpyt_lstm_cell_f32": (c_int, [_p, c_int, _p, _p, _p, _p, _p, c_int64, c_int, _p]),
    "rlpyt_atari_conv1_fwd_f32": (c_int, [_p, _p, c_int, c_int64, c_int64, _p, _p, c_float, _p, _p]),
    "rlpyt_atari_conv2_fwd_f32": (c_int, [_p, c_int64, _p, _p, _p, _p, _p]),
    "rlpyt_atari_convs_fwd_f32": (c_int, [_p, _p, c_int, c_int64, c_int64, _p, _p, _p, _p, c_float, _p,
                                          _p, _p, _p]),
    "rlpyt_atari_sample_convs_f32": (c_int, [_p, _p, c_int64, c_int64, c_int64, _p, _p, _p, _p, _p,
                                             _p, _p, _p, _p, _p, _p, c_float, _p, _p]),
    "rlpyt_atari_sample_convs_to_f32": (c_int, [_p, _p, c_int64, c_int64, c_int64, _p, _p, _p, _p, _p,
                                                _p, _p, _p, _p, _p, _p, c_float, _p, _p, _p]),
    "rlpyt_rnn_step_inputs_f32": (c_int, [_p, c_int, c_int, _p, c_int, _p, _p, _p, _p, c_int, _p, c_int, _p,
                                          _p, c_int64, _p]),
    "rlpyt_is_weights_f64": (c_int, [_p, c_int64, ctypes.c_double, _p, ctypes.c_double, _p, _p]),
    "rlpyt_lstm_seq_f32": (c_int, [_p, _p, _p, _p, _p, c_int, c_int, c_int, _p]),
    "rlpyt_lstm_seq_train_f32": (c_int, [_p, _p, _p, _p, _p, _p, _p, c_int, c_int, c_int, _p]),
    "rlpyt_lstm_seq_bwd_f32": (c_int, [_p, _p, _p, _p, _p, _p, _p, _p, _p, c_int, c_int, c_int, _p]),
    "rlpyt_dqn_convs_workspace_floats": (c_int64, [c_int64]),
    "rlpyt_dqn_convs_packed_floats": (c_int64, []),
    "rlpyt_dqn_convs_pack_f32": (c_int, [_p, _p, _p, _p, _p]),
    "rlpyt_dqn_convs_fwd_f32": (c_int, [_p, c_int64, _p, _p, _p, _p, _p, _p, _p, c_float, _p, _p, _p]),
    "rlpyt_dqn_convs_x6_packed_bytes": (c_int64, []),
    "rlpyt_dqn_convs_x6_pack": (c_int, [_p, _p, _p, _p]),
    "rlpyt_dqn_conv23_x6_f32": (c_int, [_p, c_int64, _p, _p, _p, _p, _p, _p]),
    "rlpyt_dqn_conv1_f32": (c_int, [_p, c_int64, _p, _p, c_float, _p, _p]),
    "rlpyt_dqn_convs_bwd_workspace_floats": (c_int64, [c_int64]),
    "rlpyt_dqn_convs_bwd_f32": (c_int, [_p, c_int64, _p, _p, _p, _p, _p, _p, c_float, _p, _p, _p, _p, _p, _p,
                                        _p, _p]),
    "rlpyt_q_head_f32": (c_int, [_p, c_int, _p, _p, _p, c_int64, c_int, c_int, _p, _p]),
    "rlpyt_q_head_train_f32": (c_int, [_p, c_int, _p, _p, _p, c_int64, c_int, c_int, _p, _p, _p]),
    "rlpyt_q_head_bwd_f32": (c_int, [_p, _p, _p, c_int64, c_int, c_int, _p, _p, _p, _p, _p]),
    "rlpyt_rollout_fc_ksplit": (c_int, [c_int]),
    "rlpyt_rollout_fc_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "rlpyt_rollout_fc_f32": (c_int, [_p, _p, _p, c_int, c_int, c_int, _p]),
    "rlpyt_rollout_head_f32": (c_int, [_p, c_int] + [_p] * 7 + [c_int64, c_int, c_int, _p, _p, _p,
                                                            c_int64, c_int64, _p, _p, _p]),
    "rlpyt_atari_conv_wgrad_workspace_bytes": (c_int64, []),
    "rlpyt_atari_conv2_bwd_x6_f32": (c_int, [_p, _p, _p, c_int64, _p, _p, _p, _p, _p, _p]),
    "rlpyt_atari_conv1_wgrad_f32": (c_int, [_p, _p, c_int, c_int64, c_int64, _p, c_float, _p, _p,
                                            _p, _p]),
    "rlpyt_gather_rows": (c_int, [_p, _p, _p, _p, c_int, c_int64, c_int64, c_int64, _p]),
    "rlpyt_replay_step_fields": (c_int, [_p] * 7 + [c_int64, c_int, c_int64, c_int] + [_p] * 9),
    "rlpyt_replay_append": (c_int, [_p, c_int, _p, _p, c_int64, c_int] + [c_int64] * 4 + [_p]),
    "rlpyt_frames_gather": (c_int, [_p, _p, _p, _p, _p, c_int64, c_int, c_int64, c_int, c_int64,
                                    _p]),
    "rlpyt_frames_gather_seq": (c_int, [_p, _p, _p, _p, _p, c_int64, c_int, c_int, c_int64,
                                        c_int, c_int64, _p]),
    "rlpyt_frames_gather_pair": (c_int, [_p, _p, _p, _p, _p, c_int64, c_int, c_int, c_int64,
                                         c_int, c_int64, _p]),
    "rlpyt_gather_sequences": (c_int, [_p, _p, _p, _p, c_int64, c_int, c_int, c_int64, c_int64,
                                       _p]),
    "rlpyt_clip_adam_workspace_bytes": (c_int64, []),
    "rlpyt_clip_adam_step_f32": (c_int, [_p, c_int, c_double, c_double, c_double, c_double, c_double,
                                         c_int64, c_double, _p, _p, _p]),
    "rlpyt_clip_adam_step_dev_f32": (c_int, [_p, c_int, c_double, c_double, c_double, c_double,
                                             c_double, c_int64, c_double, _p, _p, _p, _p, _p]),
    "rlpyt_clip_adam_step_mirror_f32": (c_int, [_p, c_int, c_double, c_double, c_double, c_double,
                                                c_double, c_int64, c_double, _p, _p, _p, _p, c_int, _p,
                                                c_int64, c_int64, _p]),
    "rlpyt_update_tick": (c_int, [_p, _p, c_int, c_int, _p, _p, _p, c_int64, _p, _p]),
    "rlpyt_sumtree_create": (c_int, [POINTER(c_void_p), c_int, c_int, c_int, c_int, c_double,
                                     c_int, c_int]),
    "rlpyt_sumtree_destroy": (None, [_p]),
    "rlpyt_sumtree_reset": (c_int, [_p, _p]),
    "rlpyt_sumtree_levels": (c_int, [_p]),
    "rlpyt_sumtree_low_idx": (c_int64, [_p]),
    "rlpyt_sumtree_cursor": (c_int, [_p]),
    "rlpyt_sumtree_data": (c_void_p, [_p]),
    "rlpyt_sumtree_copy_tree": (c_int, [_p, _p, _p]),
    "rlpyt_sumtree_advance": (c_int, [_p, c_int, _p, c_int, _p]),
    "rlpyt_sumtree_sample": (c_int, [_p, _p, c_int, _p, _p, _p, _p]),
    "rlpyt_sumtree_set_sampled": (c_int, [_p, _p, c_int, _p, _p]),
    "rlpyt_sumtree_update": (c_int, [_p, _p, c_int, _p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

for _name, (_res, _args) in _SIGNATURES.items():
    _fn = getattr(lib, _name)  # AttributeError here = header/library mismatch: fail loudly.
    _fn.restype = _res
    _fn.argtypes = _args

if lib.rlpyt_hip_abi_version() != ABI_VERSION:
    raise HipExtensionMissing(
        f"{LIB_PATH} has ABI {lib.rlpyt_hip_abi_version()}, binding expects {ABI_VERSION}: rebuild.")


def last_error():
    return lib.rlpyt_hip_last_error().decode("utf-8", "replace")


def check(rc, what=""):
    if rc != 0:
        raise HipError(f"{what or 'librlpyt_hip'} failed (rc={rc}): {last_error()}")


def require_gpu():
    if not torch.cuda.is_available():
        raise HipError("rlpyt_amd hot path needs an MI355X (torch.cuda.is_available() is "
                       "False); there is no CPU fallback.")


def ptr(t):
    """Device pointer of a contiguous torch tensor (or None)."""
    if t is None:
        return None
    assert t.is_cuda, "rlpyt_amd kernels take device tensors"
    assert t.is_contiguous(), "rlpyt_amd kernels take contiguous tensors"
    return c_void_p(t.data_ptr())


def stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


class _DevView:
    """Minimal __cuda_array_interface__ carrier: lets torch view a raw device address."""

    def __init__(self, ptr_, shape, typestr):
        self.__cuda_array_interface__ = dict(shape=tuple(shape), typestr=typestr,
                                             data=(int(ptr_), False), version=2, strides=None)


def host_mapped_tensor(arr, device):
    """Device tensor aliasing a page-locked (``rlpyt_host_register``-ed) numpy array: kernels
    read / write the host buffer in place over PCIe."""
    dptr = c_void_p()
    check(lib.rlpyt_host_device_pointer(c_void_p(arr.ctypes.data), ctypes.byref(dptr)),
          "rlpyt_host_device_pointer")
    assert arr.flags["C_CONTIGUOUS"]
    return torch.as_tensor(_DevView(dptr.value, arr.shape, arr.dtype.str), device=device)


def last_variant():
    """Name of the last kernel instantiation this thread launched (``rlpyt_hip_last_variant``)."""
    return lib.rlpyt_hip_last_variant().decode()


def variant_reset():
    lib.rlpyt_hip_variant_reset()


def variant_counts():
    """{kernel instantiation: launches since the last ``variant_reset``}."""
    # one pass into a buffer that is known to be large enough: sizing it with a first pass races
    # with launches from other threads (sampler serve threads), and a short buffer silently drops
    # the last-registered kernels
    cap = 1 << 16
    while True:
        buf = ctypes.create_string_buffer(cap)
        used = int(lib.rlpyt_hip_variant_dump(buf, cap))
        if used < cap:
            break
        cap = used + 4096
    out = {}
    for line in buf.value.decode().splitlines():
        name, _, cnt = line.rpartition("\t")
        out[name] = out.get(name, 0) + int(cnt)
    return out


def device_info():
    buf = ctypes.create_string_buffer(128)
    n = lib.rlpyt_hip_device_info(buf, 128)
    check(min(n, 0), "rlpyt_hip_device_info")
    return buf.value.decode(), n
