"""Shape configuration of the three hot-path stages.

Defaults are the CosyVoice2 wiring of the reference
(/root/reference/examples/tts_vc/cosyvoice2/conf/cosyvoice.yaml:17-102) and the
Qwen2.5-0.5B backbone shape the reference reads from ``config.json``
(/root/reference/cosyvoice/llm/llm.py:749-752).  ``tiny_*`` constructors give
reduced shapes used by the CPU parity tests and the golden fixtures.
"""
from dataclasses import dataclass, field, replace
from typing import Tuple


@dataclass(frozen=True)
class HiftConfig:
    # /root/reference/cosyvoice/hifigan/generator.py:228-246 (ctor defaults = v1 22.05 kHz),
    # cosyvoice2 yaml :82-102 (24 kHz, ups 8/5/3)
    in_channels: int = 80
    base_channels: int = 512
    nb_harmonics: int = 8
    sampling_rate: int = 24000
    nsf_alpha: float = 0.1
    nsf_sigma: float = 0.003
    nsf_voiced_threshold: float = 10.0
    upsample_rates: Tuple[int, ...] = (8, 5, 3)
    upsample_kernel_sizes: Tuple[int, ...] = (16, 11, 7)
    n_fft: int = 16
    hop_len: int = 4
    resblock_kernel_sizes: Tuple[int, ...] = (3, 7, 11)
    resblock_dilation_sizes: Tuple[Tuple[int, ...], ...] = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    source_resblock_kernel_sizes: Tuple[int, ...] = (7, 7, 11)
    source_resblock_dilation_sizes: Tuple[Tuple[int, ...], ...] = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    lrelu_slope: float = 0.1
    audio_limit: float = 0.99
    f0_cond_channels: int = 512

    @property
    def total_upsample(self) -> int:
        u = 1
        for r in self.upsample_rates:
            u *= r
        return u * self.hop_len

    @staticmethod
    def v2() -> "HiftConfig":
        return HiftConfig()

    @staticmethod
    def v1() -> "HiftConfig":
        # /root/reference/examples/tts_vc/cosyvoice/conf/cosyvoice.yaml:116-141
        return HiftConfig(sampling_rate=22050, upsample_rates=(8, 8), upsample_kernel_sizes=(16, 16),
                          source_resblock_kernel_sizes=(7, 11),
                          source_resblock_dilation_sizes=((1, 3, 5), (1, 3, 5)))

    @staticmethod
    def tiny() -> "HiftConfig":
        return HiftConfig(base_channels=128, upsample_rates=(4, 3), upsample_kernel_sizes=(8, 7),
                          resblock_kernel_sizes=(3, 7), resblock_dilation_sizes=((1, 3), (1, 3)),
                          source_resblock_kernel_sizes=(7, 11),
                          source_resblock_dilation_sizes=((1, 3), (1, 3)), f0_cond_channels=64)


@dataclass(frozen=True)
class FlowConfig:
    # CausalMaskedDiffWithXvec, /root/reference/cosyvoice/flow/flow.py:164-201 + yaml :35-80
    input_size: int = 512
    output_size: int = 80
    spk_embed_dim: int = 192
    vocab_size: int = 6561
    input_frame_rate: int = 25
    token_mel_ratio: int = 2
    pre_lookahead_len: int = 3
    # UpsampleConformerEncoder (transformer/upsample_encoder.py:100-235)
    enc_dim: int = 512
    enc_heads: int = 8
    enc_linear_units: int = 2048
    enc_blocks: int = 6
    enc_up_blocks: int = 4
    # ConditionalDecoder (flow/decoder.py:88-206), channels=[256]
    est_in_channels: int = 320
    est_channels: int = 256
    est_n_blocks: int = 4
    est_mid_blocks: int = 12
    est_heads: int = 8
    est_head_dim: int = 64
    est_ff_mult: int = 4
    # CFM (flow/flow_matching.py:22-36, yaml :59-67)
    inference_cfg_rate: float = 0.7
    n_timesteps: int = 10
    noise_len: int = 50 * 300

    @property
    def est_time_dim(self) -> int:
        return self.est_channels * 4

    @property
    def est_inner(self) -> int:
        return self.est_heads * self.est_head_dim

    @staticmethod
    def full() -> "FlowConfig":
        return FlowConfig()

    @staticmethod
    def tiny() -> "FlowConfig":
        # enc_dim stays 512: the reference hard-codes 512 channels in PreLookaheadLayer / Upsample1D
        # (transformer/upsample_encoder.py:207,223)
        return FlowConfig(vocab_size=200, enc_linear_units=128, enc_blocks=1, enc_up_blocks=1,
                          est_n_blocks=1, est_mid_blocks=2)


@dataclass(frozen=True)
class FlowV1Config(FlowConfig):
    """CosyVoice-v1 MaskedDiffWithXvec (/root/reference/cosyvoice/flow/flow.py:25-62, examples/tts_vc/cosyvoice/conf/
    cosyvoice.yaml:66-114): plain ConformerEncoder, InterpolateRegulator, non-causal two-level ConditionalDecoder
    channels=[256, 256] (stride-2 Downsample1D / ConvTranspose Upsample1D, GroupNorm(8) blocks)."""
    vocab_size: int = 4096
    input_frame_rate: int = 50
    reg_layers: int = 4           # len(sampling_ratios)
    reg_groups: int = 1           # InterpolateRegulator(groups=1), length_regulator.py:22-28
    est_groups: int = 8           # Block1D(groups=8), flow/components/decoder.py:31
    hop_size: int = 256           # mel hop hard-coded in flow.py:143 / length_regulator.py:55

    @staticmethod
    def full() -> "FlowV1Config":
        return FlowV1Config()

    @staticmethod
    def tiny() -> "FlowV1Config":
        return FlowV1Config(vocab_size=200, enc_linear_units=128, enc_blocks=1, est_n_blocks=1, est_mid_blocks=2)


@dataclass(frozen=True)
class LlmConfig:
    # Qwen2LM, /root/reference/cosyvoice/llm/llm.py:769-804; backbone shape = Qwen2.5-0.5B config.json
    hidden_size: int = 896
    num_layers: int = 24
    num_heads: int = 14
    num_kv_heads: int = 2
    head_dim: int = 64
    intermediate_size: int = 4864
    vocab_size: int = 151936
    rms_eps: float = 1e-6
    rope_theta: float = 1000000.0
    speech_token_size: int = 6561
    tie_word_embeddings: bool = True

    @property
    def q_dim(self) -> int:
        return self.num_heads * self.head_dim

    @property
    def kv_dim(self) -> int:
        return self.num_kv_heads * self.head_dim

    @property
    def out_vocab(self) -> int:
        return self.speech_token_size + 3

    @staticmethod
    def full() -> "LlmConfig":
        return LlmConfig()

    @staticmethod
    def tiny() -> "LlmConfig":
        return LlmConfig(hidden_size=256, num_layers=2, num_heads=4, num_kv_heads=2, head_dim=64,
                         intermediate_size=512, vocab_size=512, speech_token_size=125)

    def hf_config_dict(self) -> dict:
        """The ``config.json`` the reference's Qwen2Encoder would read (llm.py:749-752)."""
        return dict(hidden_size=self.hidden_size, num_hidden_layers=self.num_layers,
                    num_attention_heads=self.num_heads, num_key_value_heads=self.num_kv_heads,
                    intermediate_size=self.intermediate_size, vocab_size=self.vocab_size,
                    rms_norm_eps=self.rms_eps, rope_theta=self.rope_theta, hidden_act="silu",
                    max_position_embeddings=32768, tie_word_embeddings=self.tie_word_embeddings,
                    attention_dropout=0.0, use_sliding_window=False, model_type="qwen2")


__all__ = ["HiftConfig", "FlowConfig", "LlmConfig", "replace", "field"]


@dataclass(frozen=True)
class BigVGANConfig:
    # /root/reference/cosyvoice/BigVGAN/bigvgan.py:257-276 (ctor defaults); encoder1 / encoder2 are injected modules there and
    # stay injected here (None = tokens go straight to encoder_proj)
    vocab_size: int = 6561
    input_size: int = 512
    output_size: int = 1024
    mel_bin: int = 80
    upsample_rates: Tuple[int, ...] = (4, 4, 4, 4, 2, 2)
    upsample_kernel_sizes: Tuple[int, ...] = (8, 8, 4, 4, 4, 4)
    upsample_initial_channel: int = 1536
    resblock_kernel_sizes: Tuple[int, ...] = (3, 7, 11)
    resblock_dilation_sizes: Tuple[Tuple[int, ...], ...] = ((1, 3, 5), (1, 3, 5), (1, 3, 5))
    speaker_embedding_dim: int = 512
    cond_in_each_up_layer: bool = True

    @property
    def total_upsample(self) -> int:
        u = 1
        for r in self.upsample_rates:
            u *= r
        return u

    @staticmethod
    def full() -> "BigVGANConfig":
        return BigVGANConfig()

    @staticmethod
    def tiny() -> "BigVGANConfig":
        return BigVGANConfig(vocab_size=64, input_size=32, output_size=48, upsample_rates=(4, 2), upsample_kernel_sizes=(8, 4),
                             upsample_initial_channel=64, speaker_embedding_dim=16)


@dataclass(frozen=True)
class PhonemeFrontConfig:
    # Qwen2LM_Phoneme_Src2 (/root/reference/cosyvoice/llm/llm.py:1450-1531) as configured by
    # examples/tts_vc/cosyvoice2/conf/cosyvoice_pho_tts.yaml:29-73: 4-factor phoneme embedding -> ConformerEncoder
    # (rel-pos, no cnn, no macaron) -> affine -> one DecoderLayer (self-attn + src-attn over the BPE text embeddings)
    text_token_size: int = 140
    text_token_dim: int = 400
    text_tone_size: int = 16
    text_tone_dim: int = 64
    text_lang_size: int = 2
    text_lang_dim: int = 16
    text_prsd_size: int = 6
    text_prsd_dim: int = 32
    enc_dim: int = 1024
    enc_heads: int = 16
    enc_linear_units: int = 4096
    enc_blocks: int = 6
    src_heads: int = 16          # hard-coded in the reference (llm.py:1503-1505)
    src_linear_units: int = 4096
    spk_embed_dim: int = 192
    use_frontend_prsd: bool = False
    # sampling: cosyvoice.utils.common.non_random_ras_sampling (yaml :67-72)
    top_p: float = 0.8
    top_k: int = 10
    win_size: int = 10
    tau_r: float = 0.1
    expand_scale: int = 2

    @property
    def input_size(self) -> int:
        return self.text_token_dim + self.text_tone_dim + self.text_lang_dim + self.text_prsd_dim

    @staticmethod
    def full() -> "PhonemeFrontConfig":
        return PhonemeFrontConfig()

    @staticmethod
    def tiny() -> "PhonemeFrontConfig":
        return PhonemeFrontConfig(text_token_size=30, text_token_dim=24, text_tone_size=5, text_tone_dim=8, text_lang_size=2,
                                  text_lang_dim=8, text_prsd_size=4, text_prsd_dim=8, enc_dim=128, enc_heads=2,
                                  enc_linear_units=256, enc_blocks=2, spk_embed_dim=16)   # src_heads / src_linear_units are hard-coded in the reference


@dataclass(frozen=True)
class TransformerLMConfig:
    # CosyVoice-v1 TransformerLM (/root/reference/cosyvoice/llm/llm.py:41-237) as configured by
    # examples/tts_vc/cosyvoice/conf/cosyvoice.yaml:17-66
    text_token_size: int = 60515
    text_encoder_input_size: int = 512
    enc_dim: int = 1024            # text_encoder: ConformerEncoder (rel-pos, no cnn, no macaron, static_chunk_size 1 = causal)
    enc_heads: int = 16
    enc_linear_units: int = 4096
    enc_blocks: int = 6
    llm_dim: int = 1024            # llm: TransformerEncoder ('linear_legacy' input layer, rel-pos attention, ReLU FFN)
    llm_heads: int = 16
    llm_linear_units: int = 4096
    llm_blocks: int = 14
    speech_token_size: int = 4096
    spk_embed_dim: int = 192
    top_p: float = 0.8             # non_random_ras_sampling (yaml :57-62)
    top_k: int = 10
    win_size: int = 50
    tau_r: float = 0.1
    expand_scale: int = 2

    @staticmethod
    def full() -> "TransformerLMConfig":
        return TransformerLMConfig()

    @staticmethod
    def tiny() -> "TransformerLMConfig":
        return TransformerLMConfig(text_token_size=200, text_encoder_input_size=32, enc_dim=128, enc_heads=2, enc_linear_units=256,
                                   enc_blocks=2, llm_dim=128, llm_heads=2, llm_linear_units=256, llm_blocks=3, speech_token_size=96,
                                   spk_embed_dim=16)
