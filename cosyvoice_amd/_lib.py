"""ctypes binding of libcosyvoice_amd.so (the C ABI declared in include/cosyvoice_amd.h).

The product path has NO fallback: if the library is missing or a call fails, this raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcosyvoice_amd.so")

CV_F32, CV_BF16, CV_F16 = 0, 1, 2
CV_F32X3 = 3  # cv_gemm only: fp32 tensors, bf16x3 products
ACT_NONE, ACT_GELU, ACT_SILU, ACT_MISH, ACT_LEAKY, ACT_ELU, ACT_SNAKE, ACT_TANH, ACT_SWIGLU = range(9)
OUT_ROWMAJOR, OUT_QKV = 0, 1

_i32, _i64, _f32, _vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p


class GemmParams(C.Structure):
    _fields_ = [
        ("dtype", _i32), ("M", _i32), ("N", _i32), ("K", _i32), ("batch", _i32), ("batch_inner", _i32),
        ("A", _vp), ("a_bs0", _i64), ("a_bs1", _i64), ("lda", _i32), ("a_rows", _i32),
        ("cin", _i32), ("a_row_stride", _i32), ("tap_base", _i32), ("tap_step", _i32),
        ("W", _vp), ("w_bs0", _i64), ("w_bs1", _i64), ("ldw", _i32),
        ("bias", _vp),
        ("res", _vp), ("res_bs0", _i64), ("res_bs1", _i64), ("ldres", _i32),
        ("res2", _vp), ("ldres2", _i32),
        ("out_scale", _f32),
        ("act", _i32), ("act_param", _vp), ("act_slope", _f32),
        ("out_f32", _vp), ("o32_bs0", _i64), ("o32_bs1", _i64), ("ldo32", _i32),
        ("out_act", _vp), ("oa_bs0", _i64), ("oa_bs1", _i64), ("ldoa", _i32),
        ("out_row_stride", _i32), ("out_row_off", _i32), ("out_rows", _i32),
        ("out_mode", _i32),
        ("q_cols", _i32), ("k_cols", _i32), ("q_scale", _f32),
        ("k_out", _vp), ("k_bs", _i64), ("ldk", _i32),
        ("vt_out", _vp), ("vt_heads", _i32), ("vt_ld", _i32),
        ("x3_flags", _i32), ("reserved_x3", _i32),
    ]


class NormParams(C.Structure):
    _fields_ = [
        ("rows", _i32), ("dim", _i32), ("rms", _i32), ("eps", _f32),
        ("x", _vp), ("ldx", _i32),
        ("gamma", _vp), ("beta", _vp),
        ("add", _vp), ("add_ld", _i32), ("rows_per_group", _i32),
        ("act", _i32), ("out_scale", _f32), ("out_dtype", _i32),
        ("out_f32", _vp), ("ldo32", _i32),
        ("out_act", _vp), ("ldoa", _i32),
    ]


class GroupNormParams(C.Structure):
    _fields_ = [
        ("B", _i32), ("T", _i32), ("C", _i32), ("groups", _i32), ("eps", _f32),
        ("x", _vp), ("x_bs", _i64), ("ldx", _i32),
        ("gamma", _vp), ("beta", _vp),
        ("add", _vp), ("add_ld", _i32),
        ("act", _i32), ("out_dtype", _i32),
        ("out_f32", _vp), ("o32_bs", _i64), ("ldo32", _i32),
        ("out_act", _vp), ("oa_bs", _i64), ("ldoa", _i32),
        ("partial", _vp),
    ]


class AttnParams(C.Structure):
    _fields_ = [
        ("dtype", _i32), ("B", _i32), ("H", _i32), ("Hkv", _i32), ("Tq", _i32), ("Tk", _i32),
        ("q", _vp), ("q_bs", _i64), ("ldq", _i32),
        ("k", _vp), ("k_bs", _i64), ("ldk", _i32),
        ("vt", _vp), ("vt_ld", _i32),
        ("out", _vp), ("o_bs", _i64), ("ldo", _i32),
        ("scale", _f32),
        ("klen", _vp), ("chunk", _i32), ("causal", _i32), ("causal_off", _i32),
        ("bias", _vp), ("bias_bs", _i64), ("bias_hs", _i64), ("bias_ld", _i32), ("q_off", _i32),
        ("q_hs", _i64), ("k_hs", _i64),
    ]


class SkinnyParams(C.Structure):
    _fields_ = [
        ("dtype", _i32), ("M", _i32), ("N", _i32), ("K", _i32),
        ("A", _vp), ("lda", _i32),
        ("Wp", _vp), ("bias", _vp),
        ("ksplit", _i32), ("mode", _i32),
        ("out_f32", _vp), ("ldo", _i32), ("slab_stride", _i64),
        ("out_act", _vp), ("ldoa", _i32),
        ("nx", _vp), ("ldnx", _i32),
        ("nslabs", _vp), ("n_nslab", _i32), ("nslab_stride", _i64), ("ld_nslab", _i32),
        ("ngamma", _vp), ("neps", _f32),
        ("nx_out", _vp),
        ("max_wgs", _i32),
        ("xb_out", _vp), ("ldxb", _i32), ("ss_part", _vp),
        ("rs_part", _vp), ("n_rs_part", _i32), ("rs_eps", _f32),
    ]


class TBlockParams(C.Structure):
    _fields_ = [
        ("dtype", _i32), ("R", _i32), ("T", _i32), ("C", _i32), ("inner", _i32), ("ff", _i32),
        ("x", _vp), ("ldx", _i32), ("eps", _f32),
        ("g1", _vp), ("b1n", _vp), ("wqkv_p", _vp),
        ("qk", _vp), ("ldqk", _i32),
        ("vt", _vp), ("vt_ld", _i32),
        ("ao", _vp), ("ldao", _i32),
        ("wo_p", _vp), ("bo", _vp),
        ("g3", _vp), ("b3n", _vp),
        ("w1_p", _vp), ("bf1", _vp),
        ("w2_p", _vp), ("bf2", _vp),
        ("out_act", _vp), ("ldoa", _i32), ("cus", _i32),
    ]


class ResblockParams(C.Structure):
    _fields_ = [
        ("dtype", _i32), ("R", _i32), ("T", _i32), ("C", _i32), ("cin", _i32),
        ("a", _vp), ("lda", _i32),
        ("w1_p", _vp), ("b1", _vp), ("g1", _vp), ("be1", _vp), ("tadd", _vp),
        ("h1", _vp), ("ldh1", _i32),
        ("w2_p", _vp), ("b2", _vp), ("g2", _vp), ("be2", _vp),
        ("wr_p", _vp), ("br", _vp),
        ("out", _vp), ("ldo", _i32),
        ("eps", _f32), ("cus", _i32),
    ]


class SampleParams(C.Structure):
    _fields_ = [
        ("logits", _vp), ("ldl", _i32), ("V", _i32), ("B", _i32),
        ("eos", _i32), ("top_k", _i32), ("top_p", _f32), ("win_size", _i32), ("tau_r", _f32),
        ("seed", C.c_uint64),
        ("fallback_mode", _i32), ("top_p2", _f32), ("top_k2", _i32),
        ("uniforms", _vp), ("max_trials", _i32),
        ("min_len", _vp), ("max_len", _vp),
        ("forced", _vp), ("forced_ld", _i32),
        ("step", _vp), ("pos", _vp), ("n_emitted", _vp), ("finished", _vp),
        ("out_tokens", _vp), ("out_ld", _i32),
        ("emb_table", _vp), ("emb_dim", _i32),
        ("x", _vp), ("ldx", _i32),
        ("nonce", _vp),
    ]


class LlmLayer(C.Structure):
    _fields_ = [("p_qkv", _vp), ("bqkv", _vp), ("p_o", _vp), ("p_gu", _vp), ("p_down", _vp), ("g_in", _vp), ("kcache", _vp), ("vtcache", _vp)]


class LlmStepDesc(C.Structure):
    _fields_ = [
        ("dtype", _i32), ("B", _i32), ("num_layers", _i32), ("hidden", _i32), ("num_heads", _i32), ("num_kv_heads", _i32),
        ("inter", _i32), ("ctx_max", _i32), ("down_ksplit", _i32), ("rms_eps", _f32), ("split_qkv_norm", _i32), ("reserved", _i32),
        ("layers", C.POINTER(LlmLayer)),
        ("x", _vp), ("x2", _vp), ("xn", _vp), ("xb", _vp), ("ssp", _vp), ("n_ssp", _i32), ("qkv", _vp), ("ao", _vp), ("h", _vp),
        ("slabs", _vp), ("logits", _vp), ("vpad", _i32), ("rope_table", _vp), ("g_final", _vp), ("p_dec", _vp), ("dec_b", _vp),
        ("out_vocab", _i32), ("sample", SampleParams),
    ]


class FlowResnet(C.Structure):
    _fields_ = [("w1_p", _vp), ("b1", _vp), ("g1", _vp), ("be1", _vp), ("w2_p", _vp), ("b2", _vp), ("g2", _vp), ("be2", _vp),
                ("wr_p", _vp), ("br", _vp), ("cin", _i32), ("reserved", _i32)]


class FlowTBlock(C.Structure):
    _fields_ = [("g1", _vp), ("b1n", _vp), ("wqkv_p", _vp), ("wo_p", _vp), ("bo", _vp), ("g3", _vp), ("b3n", _vp),
                ("w1_p", _vp), ("bf1", _vp), ("w2_p", _vp), ("bf2", _vp)]


class FlowBlock(C.Structure):
    _fields_ = [("res", FlowResnet), ("tb", C.POINTER(FlowTBlock)), ("n_tb", _i32), ("fuse_tail_head", _i32)]


class FlowSolverDesc(C.Structure):
    _fields_ = [
        ("dtype", _i32), ("B", _i32), ("T", _i32), ("Tp", _i32),
        ("C", _i32), ("inner", _i32), ("ff", _i32), ("heads", _i32), ("in_ch", _i32), ("out_ch", _i32),
        ("n_blocks", _i32), ("n_steps", _i32), ("cus", _i32), ("cfg_rate", _f32), ("eps", _f32),
        ("blocks", C.POINTER(FlowBlock)),
        ("down_w", _vp), ("down_b", _vp), ("up_w", _vp), ("up_b", _vp),
        ("fin_w", _vp), ("fin_b", _vp), ("fin_g", _vp), ("fin_be", _vp), ("proj_w", _vp), ("proj_b", _vp),
        ("tadd", _vp), ("dts", C.POINTER(C.c_float)),
        ("x", _vp), ("mu", _vp), ("spks", _vp), ("cond", _vp), ("klen", _vp),
        ("xin", _vp), ("h1", _vp), ("x32", _vp), ("qk", _vp), ("vt", _vp), ("ao", _vp), ("cat", _vp), ("d", _vp), ("v", _vp), ("c32a", _vp),
    ]


class HiftConv(C.Structure):
    _fields_ = [("w", _vp), ("b", _vp), ("k", _i32), ("cin", _i32), ("cout", _i32), ("dilation", _i32), ("pad_left", _i32), ("stride", _i32),
                ("x3_flags", _i32), ("reserved", _i32)]


class HiftResunit(C.Structure):
    _fields_ = [("c1", HiftConv), ("c2", HiftConv), ("a1", _vp), ("a2", _vp)]


class HiftResblock(C.Structure):
    _fields_ = [("units", C.POINTER(HiftResunit)), ("n_units", _i32), ("reserved", _i32)]


class HiftPhase(C.Structure):
    _fields_ = [("w", _vp), ("ntaps", _i32), ("tap_base", _i32)]


class HiftStage(C.Structure):
    _fields_ = [("phases", C.POINTER(HiftPhase)), ("up_b", _vp), ("u", _i32), ("up_cin", _i32), ("up_flags", _i32), ("reserved", _i32),
                ("source_down", HiftConv), ("source_rb", HiftResblock), ("rbs", C.POINTER(HiftResblock)),
                ("t_out", _i32), ("c", _i32),
                ("x32", _vp), ("xa", C.POINTER(_vp)), ("r0", _vp), ("r1", _vp), ("ta", _vp), ("ra", _vp), ("acc0", _vp), ("acc1", _vp),
                ("si0", _vp), ("si1", _vp), ("out", _vp)]


class HiftDecodeDesc(C.Structure):
    _fields_ = [("dtype", _i32), ("gemm_dtype", _i32), ("B", _i32), ("T", _i32), ("S", _i32), ("n_stages", _i32), ("n_kernels", _i32),
                ("stft_ld", _i32), ("hop", _i32), ("presplit", _i32), ("lrelu_slope", _f32), ("audio_limit", _f32),
                ("conv_pre", HiftConv), ("conv_post", HiftConv), ("stages", C.POINTER(HiftStage)),
                ("mel_cl", _vp), ("s", _vp), ("stft", _vp), ("a_pre", _vp), ("post", _vp), ("wav", _vp)]


_lib = None


def lib():
    """Load the HIP library or fail loudly (no CPU fallback exists in the product path)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `python -m cosyvoice_amd.build` (hipcc, gfx950). "
                "cosyvoice_amd has no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        _lib.cv_arch.restype = C.c_char_p
        for name, st in (("gemm", GemmParams), ("norm", NormParams), ("attn", AttnParams), ("skinny", SkinnyParams),
                         ("sample", SampleParams), ("groupnorm", GroupNormParams), ("tblock", TBlockParams), ("resblock", ResblockParams)):
            fn = getattr(_lib, f"cv_sizeof_{name}_params")
            if fn() != C.sizeof(st):
                raise RuntimeError(f"ABI mismatch for cv_{name}_params: C {fn()} vs ctypes {C.sizeof(st)}")
        for name, st in (("llm_step_desc", LlmStepDesc), ("llm_layer", LlmLayer), ("flow_solver_desc", FlowSolverDesc),
                         ("flow_block", FlowBlock), ("flow_tblock", FlowTBlock), ("hift_decode_desc", HiftDecodeDesc),
                         ("hift_stage", HiftStage), ("hift_resunit", HiftResunit)):
            fn = getattr(_lib, f"cv_sizeof_{name}")
            if fn() != C.sizeof(st):
                raise RuntimeError(f"ABI mismatch for cv_{name}: C {fn()} vs ctypes {C.sizeof(st)}")
    return _lib


EXPORTS = ["cv_version", "cv_arch", "cv_gemm", "cv_layernorm", "cv_attention",
           "cv_sizeof_gemm_params", "cv_sizeof_norm_params", "cv_sizeof_attn_params",
           "cv_to_channels_last", "cv_to_channels_first", "cv_snake_multi", "cv_stft16", "cv_istft16", "cv_hift_source",
           "cv_embedding", "cv_est_pack", "cv_cfm_update", "cv_graph_begin", "cv_graph_end", "cv_graph_launch",
           "cv_graph_destroy", "cv_graph_launch_direct", "cv_graph_num_launches", "cv_stream_create_cumask",
           "cv_stream_destroy", "cv_skinny_gemm", "cv_pack_skinny", "cv_rmsnorm_reduce", "cv_rope_append",
           "cv_decode_attention", "cv_kv_retile", "cv_sample_ras", "cv_sizeof_skinny_params", "cv_sizeof_sample_params", "cv_anti_alias_act", "cv_anti_alias_act_cl",
           "cv_stft_magnitude", "cv_log_clamp_channels_first", "cv_groupnorm_cl", "cv_groupnorm_workspace_floats",
           "cv_interp_linear_cl", "cv_sizeof_groupnorm_params", "cv_relpos_append", "cv_sizeof_tblock_params", "cv_tblock_head",
           "cv_tblock_tail", "cv_tblock_tail_head", "cv_sizeof_llm_step_desc", "cv_sizeof_llm_layer", "cv_llm_step_enqueue", "cv_llm_step_graph_create",
           "cv_llm_step_graph_launch", "cv_llm_step_graph_destroy", "cv_sizeof_resblock_params", "cv_resblock_conv1", "cv_resblock_conv2",
           "cv_sizeof_flow_solver_desc", "cv_sizeof_flow_block", "cv_sizeof_flow_tblock", "cv_flow_euler_enqueue",
           "cv_flow_euler_graph_create", "cv_flow_euler_graph_launch", "cv_flow_euler_graph_destroy",
           "cv_sizeof_hift_decode_desc", "cv_sizeof_hift_stage", "cv_sizeof_hift_resunit", "cv_hift_decode_enqueue", "cv_hift_decode",
           "cv_hift_decode_graph_create"]

TORCH_DT = {torch.float32: CV_F32, torch.bfloat16: CV_BF16, torch.float16: CV_F16}
DT_TORCH = {v: k for k, v in TORCH_DT.items()}


def ptr(t):
    return None if t is None else t.data_ptr()


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed with cv_status {rc}")
