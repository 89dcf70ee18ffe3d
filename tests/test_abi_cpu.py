"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/cosyvoice_amd.h declares."""
import ctypes
import os
import re


def test_library_exports_header_symbols():
    from cosyvoice_amd import build, _lib
    path = build.build(verbose=False)
    assert os.path.exists(path)
    lib = ctypes.CDLL(path)
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "cosyvoice_amd.h")).read()
    declared = set(re.findall(r"\b(cv_[a-z0-9_]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    lib.cv_arch.restype = ctypes.c_char_p
    assert lib.cv_arch() == b"gfx950"
    _lib.lib()  # struct-size handshake


def test_exports_list_matches_header():
    """cosyvoice_amd._lib.EXPORTS (what the Python side binds) and the header declare the same entry points."""
    from cosyvoice_amd import _lib
    hdr = open(os.path.join(os.path.dirname(__file__), "..", "include", "cosyvoice_amd.h")).read()
    declared = set(re.findall(r"\b(cv_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_lib.EXPORTS), (sorted(declared - set(_lib.EXPORTS)), sorted(set(_lib.EXPORTS) - declared))


def test_argument_validation_returns_status_codes_without_a_gpu():
    """Error behaviour of the ABI: bad arguments are rejected with a negative cv_status BEFORE anything is launched (so this
    runs on the GPU-less build container), never with an exception or a crash."""
    from cosyvoice_amd import _lib as L
    lib = L.lib()
    null = ctypes.c_void_p(None)
    one = ctypes.c_void_p(16)          # a non-null, 16-byte aligned dummy that is never dereferenced on these paths
    assert lib.cv_gemm(None, null) < 0
    p = L.GemmParams()
    assert lib.cv_gemm(ctypes.byref(p), null) < 0                       # M = N = K = 0
    p.M, p.N, p.K, p.batch, p.dtype = 8, 8, 30, 1, L.CV_BF16            # K not a multiple of the 16-byte chunk
    p.A, p.W, p.out_f32, p.lda, p.ldw, p.ldo32 = 16, 16, 16, 32, 32, 8
    assert lib.cv_gemm(ctypes.byref(p), null) < 0
    p.K, p.dtype = 32, 7                                                 # unknown dtype
    assert lib.cv_gemm(ctypes.byref(p), null) < 0
    assert lib.cv_attention(None, null) < 0
    a = L.AttnParams()
    assert lib.cv_attention(ctypes.byref(a), null) < 0
    assert lib.cv_layernorm(None, null) < 0
    s = L.SkinnyParams()
    assert lib.cv_skinny_gemm(ctypes.byref(s), null) < 0
    s.M, s.N, s.K, s.Wp, s.A, s.lda, s.dtype = 17, 64, 64, 16, 16, 64, L.CV_BF16   # more than 16 rows
    assert lib.cv_skinny_gemm(ctypes.byref(s), null) < 0
    s.M, s.K = 8, 48                                                     # K not a multiple of 32
    assert lib.cv_skinny_gemm(ctypes.byref(s), null) < 0
    assert lib.cv_sample_ras(None, null) < 0
    assert lib.cv_graph_launch(null, null) < 0 and lib.cv_graph_launch_direct(null, null) < 0 and lib.cv_graph_destroy(null) < 0
    assert lib.cv_graph_num_launches(null) < 0
    assert lib.cv_stream_create_cumask(None, 8, None) < 0 and lib.cv_stream_destroy(null) < 0
    assert lib.cv_anti_alias_act(null, null, 0, 1, 1, 1, None, None, None, None, null) < 0
    assert lib.cv_anti_alias_act_cl(one, 4, 0, one, 2, 0, 1, 8, 4, None, None, None, None, null) < 0
    assert lib.cv_stft_magnitude(None, 8, None, 4, 1, 4, ctypes.c_float(1e-9), null) < 0
    assert lib.cv_log_clamp_channels_first(None, 80, None, 1, 1, 80, ctypes.c_float(1e-5), null) < 0
    assert lib.cv_pack_skinny(null, null, 16, 30, 0, null) < 0           # K % 32
    assert lib.cv_decode_attention(null, 0, null, null, None, 1, null, 0, 1, 14, 2, 704, ctypes.c_float(0.125), L.CV_BF16,
                                   None, 0, None, null) < 0
    # pre-split ([8 hi | 8 lo]) outputs need whole groups of 8 channels: C = 12 would put the last group's lo half past the row
    ptrs = (ctypes.c_void_p * 1)(16)
    assert lib.cv_snake_multi(one, 4, 12, 12, 1, ptrs, ptrs, 12, L.CV_F32X3, null) < 0
    assert lib.cv_snake_multi(one, 4, 16, 16, 1, ptrs, ptrs, 12, L.CV_F32X3, null) < 0
    assert lib.cv_tblock_head(None, null) < 0 and lib.cv_tblock_tail(None, null) < 0
    assert lib.cv_hift_decode_enqueue(None, null) < 0
