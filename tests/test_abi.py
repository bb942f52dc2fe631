"""CPU: the C-ABI library loads and exports every symbol include/gritlm_hip.h declares; argument validation
(no compute, no GPU) returns the documented error codes; the product path fails loudly without the library."""
import ctypes
import os

import pytest

from gritlm_amd import _lib


def test_library_loads_and_exports_every_header_symbol():
    lib = _lib.load()
    syms = _lib.header_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/gritlm_hip.h but not exported"
    assert set(syms) == set(_lib._SIGNATURES), "ctypes signature table out of sync with the header"
    assert lib.grit_version() == _lib.ABI_VERSION


def test_header_cites_reference_for_each_entry_point():
    src = open(_lib.HEADER_PATH).read()
    for needle in ("modeling_mistral_gritlm.py", "gritlm/gritlm.py", "training/model.py"):
        assert needle in src


def test_bad_arguments_are_rejected_without_touching_the_device():
    lib = _lib.load()
    rc = lib.grit_gemm_bf16_nt(None, None, None, 4, 16, 64, 64, 64, 16, 0, None, 0, None)
    assert rc == _lib.GRIT_E_BADARG
    assert b"null pointer" in lib.grit_last_error_string()
    buf = (ctypes.c_char * 4096)()
    p = ctypes.addressof(buf)
    p16 = (p + 15) & ~15
    assert lib.grit_gemm_bf16_nt(p16, p16, p16, 4, 16, 48, 48, 48, 16, 0, None, 0, None) == _lib.GRIT_E_UNSUPPORTED   # K % 64
    assert b"multiple of 64" in lib.grit_last_error_string()
    assert lib.grit_gemm_bf16_nt(p16, p16, p16, 4, 24, 64, 64, 64, 24, 0, None, 0, None) == _lib.GRIT_E_UNSUPPORTED   # N % 16
    assert lib.grit_gemm_bf16_nt(p16 + 2, p16, p16, 4, 16, 64, 64, 64, 16, 0, None, 0, None) == _lib.GRIT_E_BADARG    # alignment
    assert lib.grit_gemm_bf16_nt(p16, p16, p16, 4, 16, 64, 64, 64, 16, 7, None, 0, None) == _lib.GRIT_E_BADARG        # epilogue
    pair = lambda K=64, epi=0, r=None, n2=16: lib.grit_gemm_bf16_nt_pair(p16, p16, p16, r, 4, 16, K, K, 16, 16, p16, p16, p16, r, 4, n2, K, K, n2, n2,
                                                                       K, epi, None)
    assert pair(K=48) == _lib.GRIT_E_UNSUPPORTED and pair(n2=24) == _lib.GRIT_E_UNSUPPORTED                          # K % 64, N % 16
    assert pair(epi=1) == _lib.GRIT_E_BADARG and b"residual" in lib.grit_last_error_string()                          # RESIDUAL without R
    assert pair(epi=2) == _lib.GRIT_E_BADARG and b"not available" in lib.grit_last_error_string()                     # SWIGLU pairs: no
    assert lib.grit_attn_bidir_fwd(p16, p16, p16, None, 1, 8, 2, 1, 64, 256, 128, 0.1, None) == _lib.GRIT_E_UNSUPPORTED  # head_dim
    assert lib.grit_attn_causal_window_fwd(p16, p16, p16, None, 1, 8, 2, 1, 128, 512, 256, 0.1, 0, None) == _lib.GRIT_E_BADARG   # window < 1
    assert b"window" in lib.grit_last_error_string()
    assert lib.grit_attn_causal_window_varlen_bwd(p16, p16, p16, p16, p16, p16, p16, 1, 8, 8, 2, 1, 128, 512, 256, 0.1, -3, None) == _lib.GRIT_E_BADARG
    assert lib.grit_pool_norm_fwd(p16, p16, None, p16, None, 1, 8, 64, 9, 1, None) == _lib.GRIT_E_BADARG             # pooling mode
    assert lib.grit_infonce_rows_fwd_bwd(p16, p16, 50.0, p16, p16, p16, None, None, 3, 7, 8, 0, 3, 0, 7, None) == _lib.GRIT_E_BADARG  # Np % Nq
    with pytest.raises(_lib.GritHipError):
        _lib.check(-2, "x")


def test_bad_arguments_of_the_widened_entry_points():
    """Validation of the round-1 'next' entry points (grouped / RoPE GEMM, MoE, CE, decode, kNN): documented error codes, no launch."""
    lib = _lib.load()
    buf = (ctypes.c_char * 8192)()
    p16 = (ctypes.addressof(buf) + 15) & ~15
    U, B = _lib.GRIT_E_UNSUPPORTED, _lib.GRIT_E_BADARG
    assert lib.grit_gemm_bf16_nt_rope(p16, p16, p16, 4, 192, 64, 64, 64, 192, p16, p16, None, 8, 8, 128, None) == U          # N % 128
    assert b"128" in lib.grit_last_error_string()
    assert lib.grit_gemm_bf16_nt_rope(p16, p16, p16, 4, 256, 64, 64, 64, 256, p16, p16, None, 0, 8, 128, None) == B          # neither positions nor S
    assert lib.grit_gemm_bf16_nt_grouped(p16, None, p16, p16, p16, 8, 64, 64, 64, 64, 64, 4096, 64, 1, None) == B            # RESIDUAL not offered
    # the training form of the grouped GEMM: RESIDUAL not offered, SAVE without its [gate | up] buffer, BWD with a short ldr
    assert lib.grit_gemm_bf16_nt_grouped_epi(p16, None, p16, p16, None, p16, 8, 64, 64, 64, 64, 64, 4096, 64, 0, 1, None) == B
    assert lib.grit_gemm_bf16_nt_grouped_epi(p16, None, p16, p16, None, p16, 8, 64, 64, 64, 64, 64, 4096, 32, 0, 5, None) == B
    assert b"SWIGLU_STACKED_SAVE" in lib.grit_last_error_string()
    assert lib.grit_gemm_bf16_nt_grouped_epi(p16, None, p16, p16, p16, p16, 8, 64, 64, 64, 64, 64, 4096, 128, 64, 6, None) == B
    assert lib.grit_moe_combine_bwd(p16, p16, p16, p16, p16, p16, p16, 4, 60, None) == B                                     # H % 8
    assert lib.grit_moe_router_top2(p16, p16, p16, p16, 4, 64, 5, None) == U                                                 # 5 experts
    assert lib.grit_moe_index(p16, 4, 17, p16, p16, p16, p16, None) == U                                                     # > 16 experts
    assert lib.grit_ce_fwd(p16, 8, p16, p16, p16, 4, 16, None) == B                                                          # ld < V
    assert lib.grit_gemv_bf16(p16, p16, p16, 9, 16, 64, 64, 64, 16, 0, None, 0, None) == U                                   # > 8 rows
    assert b"grit_gemm_bf16_nt" in lib.grit_last_error_string()
    assert lib.grit_gemv_bf16(p16, p16, p16, 1, 48, 64, 64, 64, 24, 2, None, 0, None) == U                                   # SWIGLU N % 32
    assert lib.grit_attn_decode(p16, p16, p16, p16, p16, p16, 1, 2, 1, 64, 256, 256, 128, 0.1, None) == U                    # head_dim
    assert lib.grit_attn_decode(p16, p16, p16, p16, p16, p16, 1, 32, 2, 128, 256, 4096, 4096, 0.1, None) == U                # 16 q heads per kv head
    assert lib.grit_knn_topk(p16, p16, 2, 100, 64, 64, 1, 2000, p16, p16, p16, None) == U                                    # k > 1024
    assert lib.grit_knn_topk(p16, p16, 2, 10, 64, 64, 1, 11, p16, p16, p16, None) == U                                       # k > N
    assert lib.grit_argmax_advance(p16, 12, 16, p16, None, None, 0, None, 1, None) == B                                      # ld < V
    assert lib.grit_knn_workspace_bytes(4, 10000, 10) > 4 * 10000 * 4


def test_ops_refuse_cpu_tensors_loudly():
    import torch
    from gritlm_amd import ops
    x = torch.zeros((4, 64), dtype=torch.bfloat16)
    with pytest.raises(_lib.GritHipError, match="no CPU fallback"):
        ops.rmsnorm(x, torch.ones(64, dtype=torch.bfloat16), 1e-5)
    with pytest.raises(NotImplementedError):
        ops.pool_norm(torch.zeros((1, 2, 8), dtype=torch.bfloat16), torch.ones((1, 2), dtype=torch.int64), "weighted_mean", True)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.GritHipError, match="not built"):
        _lib.load()


def test_product_package_never_imports_the_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for dp, _, files in os.walk(os.path.join(root, "gritlm_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dp, f)).read()
                assert "gritlm_oracle" not in src and "oracle." not in src, f"{f} references the oracle"


def test_every_public_op_is_device_guarded():
    """ops.* launch on the CURRENT device's stream: every public launcher must run under the device of its first tensor argument,
    positional or keyword."""
    import inspect
    from gritlm_amd import ops
    for name, fn in vars(ops).items():
        if name.startswith("_") or not inspect.isfunction(fn) or getattr(fn, "__module__", None) != ops.__name__ or name in ops._HOST_ONLY:
            continue
        assert getattr(fn, "_device_guarded", False), f"ops.{name} is not wrapped by the device guard"


def test_comm_entry_points_validate_without_a_gpu():
    """grit_comm_* (cross-rank gather on RCCL): argument validation needs neither a GPU nor a communicator."""
    lib = _lib.load()
    B = _lib.GRIT_E_BADARG
    h = ctypes.c_void_p()
    assert lib.grit_comm_unique_id(None) == B
    assert lib.grit_comm_init(None, 2, 0, ctypes.byref(h)) == B
    buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
    assert lib.grit_comm_init(buf, 2, 2, ctypes.byref(h)) == B                       # rank outside the world
    assert lib.grit_comm_allgather_packed(None, None, 1, None, 1, 8, None, None, None) == B
    assert b"null communicator" in lib.grit_last_error_string()
    assert lib.grit_comm_destroy(None) == 0 and lib.grit_stream_destroy(None) == 0
    assert lib.grit_stream_create_cu_mask(0, ctypes.byref(h)) == B


def test_every_tracked_profile_json_parses():
    """profiles/*.json are the summaries the round's numbers are judged from: each must be ONE valid JSON document (VERDICT r05 #13 found
    two with a stderr line / two concatenated objects in them); tools/summarize_profiles.py asserts the same after writing."""
    import glob
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "profiles", "*.json")))
    assert files
    for f in files:
        with open(f) as fh:
            json.load(fh)
