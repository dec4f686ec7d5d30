"""CPU oracle for the RWKV hot path of JoelTankard/web-rwkv-gguf.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import or execute it, and only as the checker.  The product
path (``web-rwkv-gguf_amd/``) never imports this package and fails loudly when
its HIP library is missing.

The oracle is a restatement (NumPy, plus plain C under ``oracle/c``) of the
reference's algorithm for the path named by ``BASELINE.json:north_star``; every
function cites the reference file:line it follows.

Parity status:
  * chunk scheduler  -- PINNED by the reference's own known-answer tests
    (src/runtime/infer/rnn.rs:363-569), restated in tests/test_oracle_rnn.py.
  * Q8_0 / Q4_0 decode, align_offset, type sizes -- PINNED by
    src/runtime/gguf.rs:1801-1856.
  * layer_norm / l2_norm / F16 matmul definitions -- pinned by the CPU loops
    the reference tests compare against (src/tensor/ops.rs:3399-3638).
  * K-quant decode (Q4_K/Q5_K/Q6_K), WKV7, token-shift, channel-mix,
    whole-model logits -- **PARITY UNPINNED**: the reference holds no test,
    fixture or golden vector for them and it cannot be built or run here
    (Rust + wgpu; no cargo/rustc, no Vulkan ICD).  The restatement follows the
    reference's source text line by line and is self-consistent only.
"""
