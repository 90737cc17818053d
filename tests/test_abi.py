"""The C-ABI shared library loads (no GPU needed) and exports every symbol that
include/rlpyt_hip.h declares; the ctypes binding covers exactly the same set."""
import os
import re
import subprocess

from conftest import ROOT


def header_functions():
    src = open(os.path.join(ROOT, "include", "rlpyt_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"\b(rlpyt_[a-z0-9_]+)\s*\(", src)
    return sorted(set(names))


def test_library_exports_every_declared_symbol():
    from rlpyt_amd import _lib
    declared = header_functions()
    assert len(declared) >= 25
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True,
                         text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    missing = [n for n in declared if n not in exported]
    assert not missing, f"declared in rlpyt_hip.h but not exported: {missing}"


def test_binding_matches_header():
    from rlpyt_amd import _lib
    assert sorted(_lib.EXPORTED_SYMBOLS) == header_functions()
    assert _lib.lib.rlpyt_hip_abi_version() == _lib.ABI_VERSION


def test_no_cpu_fallback_without_gpu():
    """On a CPU-only torch the product path must raise, not fall back."""
    import pytest
    import torch
    from rlpyt_amd import _lib, ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.HipError):
        ops.valid_from_done(torch.zeros(4, 2, dtype=torch.bool))


def test_product_does_not_import_oracle():
    """Nothing under rlpyt_amd/ may reference the oracle (test infrastructure only)."""
    bad = []
    for dp, _dn, fns in os.walk(os.path.join(ROOT, "rlpyt_amd")):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".cpp", ".h")):
                txt = open(os.path.join(dp, fn)).read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or \
                        "/root/reference" in txt:
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad


def test_round5_entry_points_validate_their_arguments_without_a_gpu():
    """Empty inputs return OK before any device call; null pointers and unsupported shapes are
    refused with the library's error codes and a message (no compute without a GPU here)."""
    import ctypes

    from rlpyt_amd import _lib
    lib = _lib.lib
    OK = 0
    assert lib.rlpyt_dqn_convs_fwd_f32(None, -1, *([None] * 7), 1.0, None, None, None) == -1   # RLPYT_EINVAL
    buf = (ctypes.c_float * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    # N = 0 / T = 0: nothing to do
    assert lib.rlpyt_dqn_convs_fwd_f32(None, 0, *([None] * 7), 1.0, None, None, None) == OK
    assert lib.rlpyt_lstm_seq_f32(None, None, None, None, None, 0, 4, 512, None) == OK
    # null pointers
    assert lib.rlpyt_dqn_convs_fwd_f32(None, 3, *([None] * 7), 1.0, None, None, None) != OK
    assert b"null pointer" in lib.rlpyt_hip_last_error()
    assert lib.rlpyt_lstm_seq_f32(None, None, None, None, None, 2, 4, 512, None) != OK
    assert lib.rlpyt_rnn_step_inputs_f32(None, 512, 1, None, 6, None, None, None, None, 512, None, 1040,
                                         None, None, 4, None) != OK
    assert lib.rlpyt_q_head_f32(None, 8, None, None, None, 4, 512, 6, None, None) != OK
    # unsupported shapes (pointers non-null so the shape check is what fires)
    assert lib.rlpyt_lstm_seq_f32(p, p, p, p, p, 2, 4, 384, None) != OK
    assert b"H must be 256 or 512" in lib.rlpyt_hip_last_error()
    assert lib.rlpyt_q_head_f32(p, 8, p, p, p, 4, 512, 19, p, None) != OK
    assert lib.rlpyt_q_head_f32(p, 8, p, p, p, 4, 384, 6, p, None) != OK
    # replay append in one launch (ABI 17)
    fld = (_lib.AppendField * 1)()
    assert lib.rlpyt_replay_append(None, 0, None, None, 0, 1, 0, 4, 0, 10, None) == OK      # T = 0
    assert lib.rlpyt_replay_append(None, 1, None, None, 0, 1, 2, 4, 0, 10, None) != OK
    assert b"null field table" in lib.rlpyt_hip_last_error()
    assert lib.rlpyt_replay_append(fld, 1, None, None, 0, 1, 2, 4, 0, 10, None) != OK       # null ring
    assert lib.rlpyt_replay_append(None, 0, None, None, 0, 1, 11, 4, 0, 10, None) != OK     # > one lap
    assert b"at most one lap" in lib.rlpyt_hip_last_error()
    assert lib.rlpyt_replay_append(None, 0, p, None, 8, 4, 2, 4, 0, 10, None) != OK         # obs w/o frames
    # the head inside an update (ABI 16)
    assert lib.rlpyt_q_head_train_f32(p, 8, p, p, p, 4, 384, 6, p, p, None) != OK
    assert lib.rlpyt_q_head_bwd_f32(None, p, p, 4, 512, 6, p, p, p, p, None) != OK
    assert lib.rlpyt_q_head_bwd_f32(p, p, p, 257, 512, 6, p, p, p, p, None) != OK
    assert lib.rlpyt_q_head_bwd_f32(p, p, p, 4, 500, 6, p, p, p, p, None) != OK
    assert lib.rlpyt_q_head_bwd_f32(p, p, p, 4, 512, 19, p, p, p, p, None) != OK
    assert lib.rlpyt_rnn_step_inputs_f32(p, 512, 1, p, 6, p, None, p, p, 512, p, 1000, p, p, 4, None) != OK
    assert b"Kp >= F + A + 1 + H" in lib.rlpyt_hip_last_error()
    # f32 register-order copies + the bf16 pieces of w2 / w3 (3 pieces x 2 bytes per weight)
    x6 = lib.rlpyt_dqn_convs_x6_packed_bytes()
    assert x6 == (64 * 512 + 64 * 576) * 6 * 2          # 16 x 16 tile order | 32 x 32 tile order
    assert lib.rlpyt_dqn_convs_packed_floats() == 32 * 256 + 64 * 512 + 64 * 576 + x6 // 4
    assert lib.rlpyt_dqn_convs_workspace_floats(10) == 77824 + x6 // 4 + 10 * (475 * 32 + 108 * 64)
    assert lib.rlpyt_dqn_conv23_x6_f32(None, 0, None, None, None, None, None, None) == OK
    assert lib.rlpyt_dqn_conv23_x6_f32(None, 2, None, None, None, None, None, None) != OK
    assert lib.rlpyt_dqn_convs_x6_pack(None, None, None, None) != OK
    # conv1 of the DQN stack on the bf16 pipe (ABI 14)
    assert lib.rlpyt_dqn_conv1_f32(None, 0, None, None, 1.0, None, None) == OK
    assert lib.rlpyt_dqn_conv1_f32(None, 3, None, None, 1.0, None, None) != OK
    # importance-sampling weights (ABI 13)
    assert lib.rlpyt_is_weights_f64(None, 0, 1e-6, None, 0.4, None, None) == OK
    assert lib.rlpyt_is_weights_f64(None, 5, 1e-6, None, 0.4, None, None) != OK
    assert lib.rlpyt_is_weights_f64(p, 1 << 20, 1e-6, None, 0.4, p, None) != OK
    # LSTM sequence under autograd (ABI 11)
    assert lib.rlpyt_lstm_seq_train_f32(*([None] * 7), 0, 4, 512, None) == OK
    assert lib.rlpyt_lstm_seq_train_f32(*([None] * 7), 3, 4, 512, None) != OK
    assert lib.rlpyt_lstm_seq_train_f32(*([p] * 7), 3, 4, 100, None) != OK
    assert lib.rlpyt_lstm_seq_bwd_f32(*([None] * 9), 0, 4, 512, None) == OK
    assert lib.rlpyt_lstm_seq_bwd_f32(*([None] * 9), 3, 4, 512, None) != OK
    assert b"null pointer" in lib.rlpyt_hip_last_error()
    assert lib.rlpyt_lstm_seq_bwd_f32(*([p] * 9), 3, 4, 384, None) != OK
    # backward of the same stack (ABI 10)
    assert lib.rlpyt_dqn_convs_bwd_f32(None, 0, *([None] * 6), 1.0, *([None] * 8)) == OK
    assert lib.rlpyt_dqn_convs_bwd_f32(None, -2, *([None] * 6), 1.0, *([None] * 8)) == -1
    assert lib.rlpyt_dqn_convs_bwd_f32(None, 4, *([None] * 6), 1.0, *([None] * 8)) != OK
    assert b"null pointer" in lib.rlpyt_hip_last_error()
    assert lib.rlpyt_dqn_convs_bwd_workspace_floats(0) == 0
    # packed transposed weights + dz2 + dz1 + one partial per image group and layer
    n = 128
    want = (36864 + 32768) + n * (108 * 64 + 475 * 32) + 128 * (32 * 256 + 32) + 64 * (64 * 512 + 64) \
        + 64 * (64 * 576 + 64)
    assert lib.rlpyt_dqn_convs_bwd_workspace_floats(n) == want
