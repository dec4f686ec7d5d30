"""CPU-side tests of the C-ABI libraries and the native host layer (no GPU needed).

 * every function declared in include/*.h is exported by the built library;
 * the host GGUF reader / loader (C++) agrees bit-for-bit with the oracle restatement;
 * the C++ chunk scheduler reproduces the reference's known-answer tests (rnn.rs:363-569);
 * the HIP path fails loudly without a device (no CPU fallback).
"""
import os
import re

import numpy as np
import pytest

import wrk
from oracle import gguf as ogguf
from oracle import rwkv7 as orwkv7
from oracle import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(wrk_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported():
    for header, lib, table in (("wrk_hip.h", wrk.hip, wrk.HIP_SYMBOLS), ("wrk_runtime.h", wrk.rt, wrk.RT_SYMBOLS)):
        names = _declared(header)
        assert len(names) > 20
        for n in names:
            assert hasattr(lib, n), f"{n} declared in {header} but not exported"
            assert n in table, f"{n} has no ctypes signature"
        assert sorted(table) == names
    assert wrk.hip.wrk_abi_version() == 1


def test_no_gpu_means_loud_failure(has_gpu):
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(wrk.WrkError):
        wrk.Context(0)


@pytest.fixture(scope="module")
def tiny():
    data = synth.make_v7_gguf(synth.CONFIGS["tiny"], 42, mat_override={"time_mix_value": "Q5_K", "channel_mix_value": "Q8_0"})
    return data, ogguf.GgufReader(data), wrk.GgufReader(data)


def test_host_reader_matches_oracle(tiny):
    data, oref, r = tiny
    assert r.version == oref.version and r.tensor_data_offset == oref.tensor_data_offset
    names = ["emb.weight", "head.weight", "blocks.0.ln0.weight", "blocks.1.att.x_a", "blocks.0.att.r_k", "blocks.1.att.w1",
             "blocks.0.att.key.weight", "blocks.1.att.value.weight", "blocks.1.ffn.value.weight", "blocks.0.ffn.key.weight",
             "blocks.1.att.time_maa", "blocks.0.att.ln_x.bias"]
    for n in names:
        assert r.contains(n) == oref.contains(n) is True
        assert r.shape(n) == oref.shape(n), n
        got = r.tensor_f16(n).astype(np.float32)
        _, _, want = oref.tensor(n)
        want = want.astype(np.float16).astype(np.float32)     # tensor_f16_from_reader
        assert got.shape == want.shape and np.array_equal(got, want), n
    assert not r.contains("blocks.7.att.x_r")
    with pytest.raises(wrk.WrkError):
        r.shape("nope")
    t, raw = r.raw("blocks.1.att.value.weight")
    assert t == 13 and np.array_equal(raw, oref.raw_tensor("blocks.1.att.value.weight")[1])


def test_host_info_matches_oracle(tiny):
    _, oref, r = tiny
    a, b = r.info(), orwkv7.loader_info(oref)
    assert (a.version, a.num_layer, a.num_emb, a.num_hidden, a.num_vocab, a.num_head) == (7, b.num_layer, b.num_emb, b.num_hidden, b.num_vocab, b.num_head)
    assert {"w": a.lora_w, "a": a.lora_a, "g": a.lora_g, "v": a.lora_v} == b.custom


def test_host_reader_rejects_bad_files():
    for bad in (b"XXXX" + b"\0" * 64, b"GGUF" + (9).to_bytes(4, "little") + b"\0" * 64, b"GGUF" + (3).to_bytes(4, "little") + (5).to_bytes(8, "little") + (0).to_bytes(8, "little")):
        with pytest.raises(wrk.WrkError):
            wrk.GgufReader(bad)
    # a tensor table that points outside the file must be refused before anything is uploaded
    data = bytearray(synth.make_v7_gguf(synth.CONFIGS["tiny"], 42))
    with pytest.raises(wrk.WrkError):
        wrk.GgufReader(bytes(data[: len(data) // 2]))


def _craft(tensors, pad=4096):
    """Minimal GGUF v3 with a hand-written tensor table: (name, dims, type id, offset) and `pad` zero bytes of data."""
    import struct

    def wstr(s):
        return struct.pack("<Q", len(s)) + s.encode()

    head = bytearray(struct.pack("<IIQQ", 0x46554747, 3, len(tensors), 1))
    head += wstr("general.architecture") + struct.pack("<I", 8) + wstr("rwkv7")
    for nm, dims, tid, off in tensors:
        head += wstr(nm) + struct.pack("<I", len(dims)) + b"".join(struct.pack("<Q", d) for d in dims) + struct.pack("<IQ", tid, off)
    base = (len(head) + 31) & ~31
    return bytes(head) + b"\0" * (base - len(head) + pad)


def test_host_reader_overflow_and_truncation_cases():
    """file-supplied dims / offsets are overflow-checked before any pointer is formed from them."""
    ok = _craft([("blk.0.attn_norm.weight", [64], 0, 0)])
    assert wrk.GgufReader(ok).shape("blocks.0.ln1.weight") == [64]
    bad = [
        _craft([("blk.0.attn_norm.weight", [64], 0, 2 ** 64 - 32)]),                    # offset wraps past the range check
        _craft([("blk.0.attn_norm.weight", [64], 0, 2 ** 63)]),                         # plain out of range
        _craft([("blk.0.attn_norm.weight", [2 ** 33, 2 ** 33], 0, 0)]),                 # element count overflows u64
        _craft([("blk.0.attn_norm.weight", [2 ** 62], 0, 0)]),                          # byte count overflows u64
        _craft([("blk.0.attn_norm.weight", [2048], 0, 0)], pad=4096),                   # 8 KiB of f32 in a 4 KiB data area
        _craft([("blk.0.time_mix_lerp_fused.weight", [64, 1, 1, 5], 0, 0)]),            # fused lerp tensor with 5 slices, 6 are read
    ]
    for b in bad:
        with pytest.raises(wrk.WrkError):
            wrk.GgufReader(b)


L, F, N = wrk.RNN_LAST, wrk.RNN_FULL, wrk.RNN_NONE


def _inp(lens_opts, chunk):
    return wrk.RnnInput([[i] * n for i, (n, _) in enumerate(lens_opts)], chunk, [o for _, o in lens_opts])


def test_scheduler_known_answers():
    """Data of rnn.rs:363-445 (test_run_iter) and :447-503 (test_advance)."""
    it = _inp([(139, L), (1, L), (0, F), (65, F)], 128).iter()
    assert next(it) == [(65, N), (1, L), (0, F), (62, F)]
    assert next(it) == [(60, N), (1, L), (0, F), (3, F)]
    assert next(it) == [(14, L), (1, L), (0, F), (1, F)]
    assert next(it) == [(1, L), (1, L), (0, F), (1, F)]
    assert next(it) == [(1, L), (1, L), (0, F), (1, F)]
    run = _inp([(139, L), (1, L), (0, F), (65, F)], 128)
    run.step()
    assert next(run.iter()) == [(61, N), (0, L), (0, F), (3, F)]
    assert next(_inp([(61, L), (1, L), (0, F), (3, F)], 128).iter()) == [(60, N), (1, L), (0, F), (3, F)]


def test_redirect_known_answers():
    """Data of rnn.rs:505-569 (test_redirect)."""
    h, i, o = wrk.redirect(next(_inp([(61, L), (0, L), (0, F), (3, F)], 128).iter()))
    assert h == [60, 61, 62, 63] and i == [(0, 61), (61, 61), (61, 61), (61, 64)] and o == [(0, 1), (1, 1), (1, 1), (1, 4)]
    h, i, o = wrk.redirect(next(_inp([(11, L), (8, L), (9, L), (4, L)] * 2, 32).iter()))
    assert h == [15, 31]
    assert i == [(0, 4), (4, 8), (8, 12), (12, 16), (16, 20), (20, 24), (24, 28), (28, 32)]
    assert o == [(0, 0), (0, 0), (0, 0), (0, 1), (1, 1), (1, 1), (1, 1), (1, 2)]


def test_chunk_size_rounding():
    assert _inp([(1, L)], 1).token_chunk_size == 32 and _inp([(1, L)], 33).token_chunk_size == 64


def test_llama_cpp_rwkv6_names_load_like_the_attn_spellings():
    """a llama.cpp RWKV-6 file (time_mix_lerp_*, time_mix_decay*, time_mix_w1/w2 ..., general.architecture rwkv6)
    resolves to the same safetensors names, shapes and values as the attn_* spelling, in the C++ host and in the oracle; the V7
    meaning of time_mix_w1.weight (att.w1) is untouched for rwkv7 files."""
    from oracle import rwkv6 as orwkv6
    cfg = synth.V6_CONFIGS["tiny"]
    a, b = synth.make_v6_gguf(cfg, 42), synth.make_v6_gguf(cfg, 42, names="llama")
    assert a != b
    ra, rb, oa, ob = wrk.GgufReader(a), wrk.GgufReader(b), ogguf.GgufReader(a), ogguf.GgufReader(b)
    names = ["att.time_mix_x", "att.time_mix_w", "att.time_mix_g", "att.time_mix_w1", "att.time_mix_w2", "att.time_decay", "att.time_decay_w1",
             "att.time_decay_w2", "att.time_first", "att.key.weight", "att.gate.weight", "att.output.weight", "att.ln_x.bias",
             "ffn.time_mix_k", "ffn.time_mix_r", "ffn.key.weight", "ffn.value.weight", "ffn.receptance.weight"]
    for n in names:
        for layer in (0, 1):
            full = f"blocks.{layer}.{n}"
            assert rb.contains(full) and ob.contains(full), full
            assert rb.shape(full) == ra.shape(full) == ob.shape(full) == oa.shape(full), full
            assert np.array_equal(rb.tensor_f16(full), ra.tensor_f16(full)), full
            assert np.array_equal(ob.tensor(full)[2], oa.tensor(full)[2]), full
    ia, ib = ra.info(), rb.info()
    assert (ib.version, ib.num_layer, ib.num_emb, ib.num_hidden, ib.num_vocab, ib.num_head, ib.lora_w, ib.lora_a) == \
           (ia.version, ia.num_layer, ia.num_emb, ia.num_hidden, ia.num_vocab, ia.num_head, ia.lora_w, ia.lora_a) and ib.version == 6
    mb = orwkv6.loader_info_v6(ob)
    assert mb.custom == orwkv6.loader_info_v6(oa).custom
    # in an rwkv7 file the shared spellings keep their V7 meaning (gguf.rs:1253-1271)
    v7 = wrk.GgufReader(synth.make_v7_gguf(synth.CONFIGS["tiny"], 42))
    assert v7.contains("blocks.1.att.w1") and not v7.contains("blocks.1.att.time_mix_w1")


def test_host_v6_info_and_tensors_match_oracle():
    from oracle import rwkv6
    data = synth.make_v6_gguf(synth.V6_CONFIGS["tiny"], 42)
    r, oref = wrk.GgufReader(data), ogguf.GgufReader(data)
    a, b = r.info(), rwkv6.loader_info_v6(oref)
    assert (a.version, a.num_layer, a.num_emb, a.num_hidden, a.num_vocab, a.num_head) == (6, b.num_layer, b.num_emb, b.num_hidden, b.num_vocab, b.num_head)
    assert (a.lora_w, a.lora_a) == (b.custom["time_mix"], b.custom["time_decay"])
    assert r.shape("blocks.0.att.time_mix_w2") == oref.shape("blocks.0.att.time_mix_w2") == [5, 256, 32]
    for n in ("blocks.1.att.time_mix_w2", "blocks.0.att.time_first", "blocks.1.att.gate.weight", "blocks.0.ffn.time_mix_r"):
        want = oref.tensor(n)[2].astype(np.float16).astype(np.float32)
        assert np.array_equal(r.tensor_f16(n).astype(np.float32), want), n


@pytest.mark.parametrize("kind", ["Q2_K", "Q3_K", "Q4_0"])
def test_host_load_time_dequant_kinds(kind):
    """Load-time-only kinds (gguf.rs:42-75, 280-423): the C++ reader's f16 output is bit-identical to the oracle's."""
    data = synth.make_v7_gguf(synth.CONFIGS["tiny"], 3, mat=kind)
    oref, r = ogguf.GgufReader(data), wrk.GgufReader(data)
    for n in ("blocks.0.att.key.weight", "blocks.1.ffn.value.weight", "blocks.1.att.output.weight"):
        got = r.tensor_f16(n)
        _, _, want = oref.tensor(n)
        assert np.array_equal(got.view(np.uint16), want.astype(np.float16).view(np.uint16)), (kind, n)


def test_read_state_matches_oracle():
    """read_state (v7.rs:1229-1262): a state-tuned file's `blocks.N.att.time_state` [H, S, S] lands in rows 1..S of the
    state, channel h*S + c of row 1 + j = time_state[h][j][c]; the C++ reader and the oracle agree bit for bit."""
    from oracle.gguf import write_gguf
    from oracle.quantize import QUANTIZE
    cfg = synth.CONFIGS["tiny"]
    H, S, D, L = cfg.num_emb // cfg.head_size, cfg.head_size, cfg.num_emb, cfg.num_layer
    base = ogguf.GgufReader(synth.make_v7_gguf(cfg, 5))
    tensors = []
    for g in base.tensors.values():
        tensors.append((g.name, list(g.dimensions), g.type_name, np.frombuffer(base.get_tensor_data(g), np.uint8)))
    rng = np.random.default_rng(3)
    ts = [(rng.standard_normal((H, S, S)) * 0.1).astype(np.float32) for _ in range(L)]
    for l in range(L):
        tensors.append((f"blk.{l}.attn_time_state", [S, S, H], "F16", QUANTIZE["F16"](ts[l])))
    meta = [("general.architecture", "str", "rwkv7"), ("general.alignment", "u32", 32), ("rwkv7.block_count", "u32", L),
            ("rwkv7.embedding_length", "u32", D), ("rwkv7.wkv.head_size", "u32", S)]
    data = write_gguf(meta, tensors)
    want = orwkv7.read_state(ogguf.GgufReader(data))
    got = wrk.GgufReader(data).read_state()
    assert got.shape == want.shape == (L, S + 2, D)
    assert np.array_equal(got, want)
    assert np.all(got[:, 0] == 0) and np.all(got[:, S + 1] == 0)
    l, h, j, c = 1, 2, 5, 7
    assert got[l, 1 + j, h * S + c] == np.float32(np.float16(ts[l][h, j, c]))
    with pytest.raises(wrk.WrkError):
        wrk.GgufReader(synth.make_v7_gguf(cfg, 5)).read_state()      # no time_state tensors in a plain model


def test_quantile_student_levels():
    """Float4Quant::new_student (matrix.rs:29-44): 16 monotone levels in [-1, 1], 0 at index 7, equal to the oracle's (scipy) values."""
    from oracle import wrkquant
    for nu in (5.0, 3.0, 30.0):
        got, want = wrk.quantile_student(nu), wrkquant.quantile_student(nu)
        assert got.shape == (16,) and got[7] == 0.0 and got[15] == 1.0 and np.all(np.diff(got) > 0)
        assert abs(got[0] + 1.0) < 1e-6                              # p[0] = delta and p[15] = 1 - delta are symmetric
        assert np.allclose(got, want, rtol=1e-6, atol=1e-7), (nu, np.abs(got - want).max())
    with pytest.raises(wrk.WrkError):
        wrk.quantile_student(-1.0)


# ------------------------------------------------------------------ INTEGRATION.md's Rust binding vs include/wrk_hip.h
def _load_gen():
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_rust_binding", os.path.join(ROOT, "tools", "gen_rust_binding.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_integration_rust_binding_matches_the_header():
    """The `extern "C"` block a maintainer would paste into src/backend/hip.rs is generated from the header; the copy in INTEGRATION.md
    must be that output, byte for byte."""
    gen = _load_gen()
    want = gen.generate(open(gen.HEADER).read())
    got = gen.doc_block(open(gen.DOC).read())
    assert got is not None, "INTEGRATION.md has no generated Rust block"
    assert got == want, "INTEGRATION.md is stale: run `python tools/gen_rust_binding.py --write`"


def test_integration_rust_binding_independent_check():
    """Independent of the generator: parse the Rust block and the C header separately; every export is bound, with the same arity and
    the same pointer / scalar kind and pointee per argument (a `*const WrkBuf` where the header takes `const wrk_tensor*` is UB)."""
    import re
    gen = _load_gen()
    rust = gen.doc_block(open(gen.DOC).read())
    hdr = re.sub(r"/\*.*?\*/", "", open(gen.HEADER).read(), flags=re.S)
    hdr = re.sub(r"//[^\n]*", "", hdr)
    cfun = {}
    for m in re.finditer(r"^\s*([\w\s\*]+?)\s*\b(wrk_\w+)\s*\(([^;{]*?)\)\s*;", hdr, flags=re.M | re.S):
        args = " ".join(m.group(3).split())
        cfun[m.group(2)] = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
    rfun = {}
    for m in re.finditer(r"pub fn (wrk_\w+)\((.*?)\)(?: -> [^;]+)?;", rust):
        rfun[m.group(1)] = [a.split(":", 1)[1].strip() for a in m.group(2).split(",")] if m.group(2).strip() else []
    assert set(cfun) == set(rfun), (sorted(set(cfun) - set(rfun)), sorted(set(rfun) - set(cfun)))
    scal = {"int32_t": "i32", "uint32_t": "u32", "uint16_t": "u16", "size_t": "usize", "float": "f32", "void": "c_void", "char": "c_char", "int": "i32"}

    def camel(c):
        return "".join(p.capitalize() for p in c.split("_"))
    for name, cargs in cfun.items():
        rargs = rfun[name]
        assert len(cargs) == len(rargs), name
        for ca, ra in zip(cargs, rargs):
            depth = ca.count("*")
            base = re.sub(r"\bconst\b", "", ca).replace("*", " ").split()[0]
            assert ra.count("*") == depth, (name, ca, ra)
            pointee = ra.replace("*const", "").replace("*mut", "").strip()
            assert pointee == (scal.get(base) or camel(base)), (name, ca, ra)
            if depth == 1:
                assert ra.startswith("*const") == bool(re.search(r"\bconst\b", ca)), (name, ca, ra)
