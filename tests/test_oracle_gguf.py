"""GGUF reader / dequantiser oracle.

Pins from the reference's own tests (src/runtime/gguf.rs:1801-1856): type sizes, the Q8_0 block
(scale 1, values 0..31), the Q4_0 block of 0x88 bytes, align_offset.  K-quants have no reference
test ("parity unpinned"): they are checked for self-consistency against an independent scalar
restatement written straight from the block layouts in gguf.rs:95-274.
"""
import struct

import numpy as np
import pytest

from oracle import dequant as dq
from oracle import gguf, quantize as qz, rwkv7, synth


def test_type_sizes():
    assert dq.BLOCK_BYTES["F32"] == 4 and dq.BLOCK_BYTES["F16"] == 2
    assert dq.BLOCK_BYTES["Q8_0"] == 34 and dq.BLOCK_BYTES["Q4_0"] == 18
    assert dq.BLOCK_BYTES["Q4_K"] == 144 and dq.BLOCK_BYTES["Q5_K"] == 176 and dq.BLOCK_BYTES["Q6_K"] == 210


def test_dequantize_q8_0_reference_vector():
    block = np.frombuffer(np.float16(1.0).tobytes() + bytes(range(32)), dtype=np.uint8)
    out = dq.dequantize_q8_0(block, 32)
    assert out.shape == (32,)
    assert np.array_equal(out, np.arange(32, dtype=np.float32))


def test_dequantize_q4_0_reference_vector():
    block = np.frombuffer(np.float16(1.0).tobytes() + bytes([0x88] * 16), dtype=np.uint8)
    out = dq.dequantize_q4_0(block, 32)
    assert out.shape == (32,) and np.all(out == 0.0)


def test_align_offset():
    assert [gguf.align_offset(o, 32) for o in (0, 1, 32, 33)] == [0, 32, 32, 64]


def _h(b, o):
    return float(np.frombuffer(bytes(b[o:o + 2]), dtype="<f2")[0])


def _scalar_q4k(block):
    d, dmin = np.float32(_h(block, 0)), np.float32(_h(block, 2))
    sc = block[4:16]
    qs = block[16:144]
    out = []

    def sm(j):
        if j < 4:
            return sc[j] & 63, sc[j + 4] & 63
        return (sc[j + 4] & 0xF) | ((sc[j - 4] >> 6) << 4), (sc[j + 4] >> 4) | ((sc[j] >> 6) << 4)

    for g in range(4):
        s0, m0 = sm(2 * g)
        s1, m1 = sm(2 * g + 1)
        for l in range(32):
            out.append(np.float32(d * np.float32(s0)) * np.float32(qs[32 * g + l] & 0xF) - np.float32(dmin * np.float32(m0)))
        for l in range(32):
            out.append(np.float32(d * np.float32(s1)) * np.float32(qs[32 * g + l] >> 4) - np.float32(dmin * np.float32(m1)))
    return np.array(out, dtype=np.float32)


def _scalar_q6k(block):
    ql, qh = block[0:128], block[128:192]
    sc = np.frombuffer(bytes(block[192:208]), dtype=np.int8)
    d = np.float32(_h(block, 208))
    out = np.zeros(256, np.float32)
    for n in range(2):
        for l in range(32):
            i = l // 16
            a = int(ql[64 * n + l]); b = int(ql[64 * n + l + 32]); h = int(qh[32 * n + l])
            qv = [((a & 0xF) | ((h & 3) << 4)) - 32, ((b & 0xF) | (((h >> 2) & 3) << 4)) - 32,
                  ((a >> 4) | (((h >> 4) & 3) << 4)) - 32, ((b >> 4) | (((h >> 6) & 3) << 4)) - 32]
            for k in range(4):
                out[128 * n + 32 * k + l] = np.float32(d * np.float32(sc[8 * n + i + 2 * k])) * np.float32(qv[k])
    return out


def _scalar_q5k(block):
    d, dmin = np.float32(_h(block, 0)), np.float32(_h(block, 2))
    sc = block[4:16]; qh = block[16:48]; ql = block[48:176]

    def sm(j):
        if j < 4:
            return sc[j] & 63, sc[j + 4] & 63
        return (sc[j + 4] & 0xF) | ((sc[j - 4] >> 6) << 4), (sc[j + 4] >> 4) | ((sc[j] >> 6) << 4)

    out = []
    u1, u2 = 1, 2
    for g in range(4):
        s0, m0 = sm(2 * g); s1, m1 = sm(2 * g + 1)
        for l in range(32):
            q = (int(ql[32 * g + l]) & 0xF) + (16 if qh[l] & u1 else 0)
            out.append(np.float32(d * np.float32(s0)) * np.float32(q) - np.float32(dmin * np.float32(m0)))
        for l in range(32):
            q = (int(ql[32 * g + l]) >> 4) + (16 if qh[l] & u2 else 0)
            out.append(np.float32(d * np.float32(s1)) * np.float32(q) - np.float32(dmin * np.float32(m1)))
        u1 <<= 2; u2 <<= 2
    return np.array(out, dtype=np.float32)


@pytest.mark.parametrize("tn,scalar", [("Q4_K", _scalar_q4k), ("Q5_K", _scalar_q5k), ("Q6_K", _scalar_q6k)])
def test_kquant_vectorised_matches_scalar(tn, scalar):
    rng = np.random.default_rng(7)
    nb = 6
    raw = rng.integers(0, 256, nb * dq.BLOCK_BYTES[tn], dtype=np.uint8).reshape(nb, -1)
    # keep the f16 scale fields finite
    for off in ((0, 2) if tn != "Q6_K" else (208,)):
        raw[:, off:off + 2] = np.frombuffer(rng.uniform(1e-3, 5e-2, nb).astype("<f2").tobytes(), np.uint8).reshape(nb, 2)
    got = dq.DEQUANT[tn](raw.reshape(-1), nb * 256, round_f16=False).reshape(nb, 256)
    for b in range(nb):
        assert np.array_equal(got[b], scalar(raw[b])), tn
    got16 = dq.DEQUANT[tn](raw.reshape(-1), nb * 256, round_f16=True)
    assert np.array_equal(got16, got.reshape(-1).astype(np.float16).astype(np.float32))


@pytest.mark.parametrize("tn", ["Q4_K", "Q5_K", "Q6_K", "Q8_0", "Q4_0"])
def test_quantise_roundtrip_is_close(tn):
    w = synth.normal(3, "w", 256 * 16) * np.float32(0.05)
    raw = qz.QUANTIZE[tn](w)
    assert raw.size == dq.data_size(tn, w.size)
    back = dq.DEQUANT[tn](raw, w.size, round_f16=False)
    tol = {"Q4_K": 0.02, "Q5_K": 0.01, "Q6_K": 0.006, "Q8_0": 0.002, "Q4_0": 0.03}[tn]
    assert np.abs(back - w).max() < tol


def test_scale_min_pack_roundtrip():
    rng = np.random.default_rng(1)
    sc = rng.integers(0, 64, (5, 8), dtype=np.uint8)
    m = rng.integers(0, 64, (5, 8), dtype=np.uint8)
    s2, m2 = dq.get_scale_min_k4(qz.pack_scale_min_k4(sc, m))
    assert np.array_equal(sc, s2) and np.array_equal(m, m2)


def test_reader_name_map_shapes_and_fused_slices():
    cfg = synth.CONFIGS["tiny"]
    data = synth.make_v7_gguf(cfg, 42)
    r = gguf.GgufReader(data)
    assert r.version == 3 and r.tensor_data_offset % 32 == 0
    assert r.shape("emb.weight") == [cfg.num_vocab, cfg.num_emb]            # reversed dims
    assert r.shape("blocks.0.ffn.key.weight") == [cfg.num_hidden, cfg.num_emb]
    assert r.shape("blocks.0.att.r_k") == [cfg.num_head, cfg.head_size]      # r_k reshape
    assert r.shape("blocks.1.att.x_k") == [cfg.num_emb]                      # virtual slice
    assert r.contains("blocks.1.att.x_g") and not r.contains("blocks.9.att.x_g")
    fused = r.tensor("blocks.1.att.time_maa")[2].reshape(6, cfg.num_emb)
    for i, n in enumerate(["x_r", "x_w", "x_k", "x_v", "x_a", "x_g"]):
        dt, shape, vals = r.tensor(f"blocks.1.att.{n}")
        assert shape == [cfg.num_emb, 1] and np.array_equal(vals, fused[i])
    # quantized_tensor gate: K-quants are not handed out raw at HEAD (F1)
    assert r.quantized_tensor("blocks.0.att.key.weight") is None
    assert r.raw_tensor("blocks.0.att.key.weight")[0] == "Q4_K"
    info = rwkv7.loader_info(r)
    assert (info.num_layer, info.num_emb, info.num_hidden, info.num_vocab, info.num_head) == (2, 256, 1024, 512, 4)
    assert info.custom == {"w": 32, "a": 32, "g": 64, "v": 32}


def test_reader_errors():
    with pytest.raises(gguf.GgufError):
        gguf.GgufReader(b"XXXX" + b"\0" * 32)
    with pytest.raises(gguf.GgufError):
        gguf.GgufReader(struct.pack("<II", gguf.GGUF_MAGIC, 7) + b"\0" * 32)
    with pytest.raises(gguf.GgufError):
        gguf.GgufReader(struct.pack("<IIQQ", gguf.GGUF_MAGIC, 3, 1, 0))   # truncated tensor table


def test_synth_is_bit_stable():
    import hashlib
    data = synth.make_v7_gguf(synth.CONFIGS["tiny"], 42)
    # pinned when the fixture generator was written; a change means fixtures must be regenerated
    assert hashlib.sha256(data).hexdigest() == open(__file__.replace("test_oracle_gguf.py", "golden/tiny_q4k.sha256")).read().strip()


def test_q2_k_known_answer():
    """Hand-built Q2_K block (gguf.rs:372-423): scales byte 0x21 -> scale 1, min 2; d = 0.5, dmin = 0.25."""
    import numpy as np
    from oracle import dequant as dq
    blk = np.zeros(84, np.uint8)
    blk[0:16] = 0x21
    blk[16:80] = 0b11100100                      # bit pairs (LSB first): 0, 1, 2, 3 for planes j = 0..3
    blk[80:82] = np.array([0.5], "<f2").view(np.uint8)
    blk[82:84] = np.array([0.25], "<f2").view(np.uint8)
    out = dq.dequantize("Q2_K", blk, 256, round_f16=True).reshape(2, 4, 32)
    for j in range(4):
        assert np.all(out[:, j] == 0.5 * 1 * j - 0.25 * 2)


def test_q3_k_known_answer():
    """Hand-built Q3_K block (gguf.rs:280-366): every 6-bit scale = 33 (-> +1), d = 0.25, q2 plane j = j, high bits
    set for the first 128 elements only (-> no -4 there, -4 in the second half)."""
    import numpy as np
    from oracle import dequant as dq
    blk = np.zeros(110, np.uint8)
    blk[0:32] = 0x0F                             # hmask bits 0..3 set: planes of n = 0
    blk[32:96] = 0b11100100
    # scales: low 4 bits = 1 everywhere, high 2 bits = 2 everywhere  (33 = 0b100001)
    blk[96:104] = 0x11
    blk[104:108] = 0b10101010
    blk[108:110] = np.array([0.25], "<f2").view(np.uint8)
    out = dq.dequantize("Q3_K", blk, 256, round_f16=True).reshape(2, 4, 32)
    for j in range(4):
        assert np.all(out[0, j] == 0.25 * 1 * j)
        assert np.all(out[1, j] == 0.25 * 1 * (j - 4))


def test_q2_k_q3_k_fixture_roundtrip():
    import numpy as np
    from oracle import dequant as dq, quantize as qz
    w = (np.random.default_rng(5).standard_normal((4, 512)) * 0.05).astype(np.float32)
    for kind, tol in (("Q2_K", 0.08), ("Q3_K", 0.06)):
        raw = qz.QUANTIZE[kind](w)
        assert raw.size == w.size // 256 * dq.BLOCK_BYTES[kind]
        d = dq.dequantize(kind, raw, w.size, round_f16=False).reshape(w.shape)
        assert np.abs(d - w).max() < tol, kind
