"""The persistent batch-1 decode engine (web-rwkv-gguf_amd/csrc/wrk_v7_engine.hip: all layers of a token in one launch) against the
five-launch layer it replaces.

The engine keeps the thread mapping and the rounding points of the launches (with one workgroup per head: WRK_SPLIT_HEAD=0), so the
comparison is EXACT: logits, greedy tokens and the recurrent state must be bit-identical.  Parity against the oracle comes through the
launch path (tests/test_gpu_model.py, test_gpu_layer_parity.py, test_gpu_fullsize_oracle.py), which the same tests also run with the engine
on (mode 1, one sequence) wherever the model fits it.
"""
import os
import sys

import numpy as np
import pytest

import wrk
from oracle import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = wrk.Context(0)
    yield c
    c.close()


def forced(rt, toks, batch, mode=1):
    out = []
    for t in toks:
        out.append(rt.infer_raw([t], [batch | (0 << 8) | (1 << 24)], [0], mode=mode)[0])
    return np.stack(out)


def run_both(monkeypatch, rt, fn):
    """fn(rt) with the engine, then with the launches (one workgroup per head), from the same zero state."""
    res = []
    for engine in ("1", "0"):
        monkeypatch.setenv("WRK_ENGINE", engine)
        monkeypatch.setenv("WRK_SPLIT_HEAD", "0")
        z = np.zeros_like(rt.state_back(0))
        for b in range(rt.num_batch):
            rt.state_load(z, b)
        res.append(fn(rt))
    return res


@pytest.mark.parametrize("name,weights", [("tiny", wrk.WEIGHTS_INLINE), ("tiny", wrk.WEIGHTS_INLINE_F16), ("small", wrk.WEIGHTS_INLINE),
                                          ("small", wrk.WEIGHTS_INLINE_F16)])
def test_engine_equals_launches_small_models(ctx, monkeypatch, name, weights):
    data = synth.make_v7_gguf(synth.CONFIGS[name], 42)
    rt = wrk.Runtime(ctx, wrk.GgufReader(data), num_batch=2, weights=weights)
    rt.num_batch = 2
    try:
        monkeypatch.setenv("WRK_ENGINE", "1")
        ok, why = rt.engine_status()
        assert ok, why
        V = rt.info.num_vocab
        toks = [(5 + 37 * i) % (V - 1) for i in range(14)]

        def job(r):
            lg = forced(r, toks, 1)                      # sequence slot 1: the batch offset of the state is exercised
            return lg, r.state_back(1), r.state_back(0)
        (a, sa, s0a), (b, sb, s0b) = run_both(monkeypatch, rt, job)
        assert np.isfinite(a).all()
        assert np.array_equal(a, b), float(np.abs(a - b).max())
        assert np.array_equal(sa, sb)
        assert np.array_equal(s0a, s0b) and not s0a.any()     # the other slot is untouched
        assert np.abs(sa).max() > 0
    finally:
        rt.close()


def test_engine_greedy_loop_and_layer_entry(ctx, monkeypatch):
    data = synth.make_v7_gguf(synth.CONFIGS["small"], 42)
    rt = wrk.Runtime(ctx, wrk.GgufReader(data), num_batch=1, weights=wrk.WEIGHTS_INLINE)
    rt.num_batch = 1
    try:
        (ta, la), (tb, lb) = run_both(monkeypatch, rt, lambda r: (lambda o: (o[0], o[2]))(r.generate_greedy([3], 40, mode=1, want_logits=True)))
        assert np.array_equal(ta, tb)
        assert np.array_equal(la, lb)
        # teacher-forced single layers (layer 0 writes the layer-0 value, layer 2 reads it)
        D = rt.info.num_emb
        rng = np.random.default_rng(3)
        x = rng.standard_normal((1, D)).astype(np.float16)
        vf = rng.standard_normal((1, D)).astype(np.float16)

        monkeypatch.setenv("WRK_ENGINE_INSPECT", "1")         # the layer entry point runs the launches (it serves frame read-back) unless told otherwise

        def layers(r):
            outs = []
            for layer, v in ((0, None), (2, vf)):
                r.infer_layer(layer, x, v, [0 | (0 << 8) | (1 << 24)], mode=1)
                outs.append(r.frame("x", 1).copy())
                if layer == 0:
                    outs.append(r.frame("att_v0", 1).copy())
            outs.append(r.state_back(0))
            return outs
        ea, eb = run_both(monkeypatch, rt, layers)
        for u, v in zip(ea, eb):
            assert np.array_equal(u, v)
    finally:
        rt.close()


def test_engine_equals_launches_headline_model(ctx, monkeypatch):
    """bench.py's 1.5B Q4_K_M model (the configuration the metric is quoted on): 12 teacher-forced tokens + a 24-token greedy run."""
    gg = bench.make_model_gguf("1.5B", seed=7)
    rt = wrk.Runtime(ctx, wrk.GgufReader(gg), num_batch=1, weights=wrk.WEIGHTS_INLINE)
    rt.num_batch = 1
    try:
        monkeypatch.setenv("WRK_ENGINE", "1")
        ok, why = rt.engine_status()
        assert ok, why
        toks = [17 + 977 * i for i in range(12)]
        (a, sa), (b, sb) = run_both(monkeypatch, rt, lambda r: (forced(r, toks, 0), r.state_back(0)))
        assert np.isfinite(a).all() and a.std() > 0.1
        assert np.array_equal(a, b), float(np.abs(a - b).max())
        assert np.array_equal(sa, sb)
        (ta, _), (tb, _) = run_both(monkeypatch, rt, lambda r: r.generate_greedy([17], 24, mode=1))
        assert np.array_equal(ta, tb)
    finally:
        rt.close()
