"""CPU oracle — TEST INFRASTRUCTURE ONLY.

Plain PyTorch fp32 (CPU) restatements of the reference's hot-path arithmetic
(duj12/CosyVoice @ /root/reference), written from reading the reference and pinned
against the reference itself through the golden fixtures in ``tests/golden`` (minted
by ``tests/golden/make_golden.py``, which imports the reference in the build
container).  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package; the product path (``cosyvoice_amd``)
never does and fails loudly when its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * HiFT, F0 predictor, flow encoder, CFM solver, Qwen2 step: pinned against the
    reference's own modules run in the build container (tests/golden/*.npz).
  * Flow estimator: reference code imported with a restatement of three absent
    third-party classes (diffusers 0.27.2 Attention / GELU / LoRACompatibleLinear)
    -> "parity unpinned" at that third-party boundary.
  * Samplers: semantics of utils/common.py:109-146 with injected uniforms; the
    reference's RNG stream (torch.multinomial) is not reproducible -> distributional parity.
"""
