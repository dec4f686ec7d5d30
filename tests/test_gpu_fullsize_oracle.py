"""Full-size parity of the headline configuration against the oracle.

The bench's RWKV-7 1.5B Q4_K_M model (`bench.make_model_gguf("1.5B")`: 24 layers, D = 2048, Q4_K matrices, Q6_K head,
F16 LoRA) and the 2.9B shape (K = 2560) run on the HIP path with `WEIGHTS_INLINE_F16` -- the arithmetic
`oracle/c/wrk_oracle.c` restates: every decoded weight rounded to f16 (the reference dequantises K-quants to f16 at load,
gguf.rs:95-274), f16 activation buffers, f32 accumulation, op order v7.rs:716-1036 -- and are compared with that C oracle
token by token, TEACHER-FORCED (both sides are fed the same token, so one differing arg-max cannot fork the comparison).

Checked per step: the greedy token (arg-max) is identical; logit error max / mean / fraction within 1e-3 (north_star's bar)
are recorded in gpurun_out/fullsize_parity.json and bounded below.  Also: a 96-token prefill chunk (MFMA tile GEMM +
chunk WKV) against the oracle fed the same 96 tokens one by one (a chunk of one sequence IS that recurrence), and the
final recurrent state.

What the bars mean: both sides compute the same f16-rounded products; they differ in f32 summation order only.  A last-bit
f32 difference can move an f16 store by one ulp and 24 layers of a random-weight model propagate it, so the whole-model
difference is NOT bounded by 1e-3 in max-norm; tests/test_gpu_layer_parity.py bounds the per-layer
error without propagation, tests/test_gpu_f32_activations.py shows the 1e-3 bar met when no f16 store exists.
PARITY UNPINNED against the real reference (it cannot run here); the oracle is its line-by-line restatement.
"""
import json
import os
import sys

import numpy as np
import pytest

import wrk

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from oracle import cport  # noqa: E402

pytestmark = pytest.mark.gpu
OUT = os.path.join(ROOT, "gpurun_out", "fullsize_parity.json")

# bars on |logit_hip - logit_oracle| over all V logits of a step (logits have unit variance on these models)
TOL_MAX, TOL_MEAN, MIN_FRAC_1E3 = 6e-2, 8e-3, 0.10


def record(key, stats):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    data = {}
    if os.path.exists(OUT):
        try:
            data = json.load(open(OUT))
        except Exception:
            data = {}
    data[key] = stats
    json.dump(data, open(OUT, "w"), indent=1, sort_keys=True)


def stats_of(diffs):
    d = np.abs(np.asarray(diffs, np.float64))
    return {"max": float(d.max()), "mean": float(d.mean()), "frac_le_1e-3": float((d <= 1e-3).mean()),
            "frac_le_1e-2": float((d <= 1e-2).mean()), "steps": int(d.shape[0])}


class Rig:
    def __init__(self, name, seed):
        cport.lib.orc_set_threads(cport.usable_cpus())
        self.gg = bench.make_model_gguf(name, seed=seed)
        self.ctx = wrk.Context(0)
        self.rt = wrk.Runtime(self.ctx, wrk.GgufReader(self.gg), num_batch=1, weights=wrk.WEIGHTS_INLINE_F16)
        self.oracle = cport.CModel(self.gg)
        self.V = self.rt.info.num_vocab

    def zero(self):
        self.rt.state_load(np.zeros_like(self.rt.state_back(0)), 0)
        self.oracle.state[:] = 0

    def close(self):
        self.rt.close()
        self.ctx.close()


@pytest.fixture(scope="module")
def rig15():
    r = Rig("1.5B", 7)
    yield r
    r.close()


def forced_decode(rig, tokens, mode):
    """Feed the same tokens to both sides, one decode step each; returns (hip logits, oracle logits) [steps, V]."""
    rig.zero()
    got, want = [], []
    for t in tokens:
        got.append(rig.rt.infer_raw([t], [0 | (0 << 8) | (1 << 24)], [0], mode=mode)[0].copy())
        want.append(rig.oracle.decode(int(t)).copy())
    return np.stack(got), np.stack(want)


def check(rig, key, got, want):
    st = stats_of(got - want)
    st["greedy_identical"] = bool((got.argmax(axis=1) == want.argmax(axis=1)).all())
    # margin between the two best logits of the oracle, smallest over the steps: the arg-max is only DEFINED up to the noise
    top2 = np.sort(want, axis=1)[:, -2:]
    st["min_top1_margin"] = float((top2[:, 1] - top2[:, 0]).min())
    record(key, st)
    print(key, st)
    assert st["greedy_identical"], st
    assert st["max"] <= TOL_MAX and st["mean"] <= TOL_MEAN and st["frac_le_1e-3"] >= MIN_FRAC_1E3, st
    return st


TOKENS = [(17 + 977 * i) % 65535 for i in range(16)]


@pytest.mark.parametrize("mode", [1, 0])
def test_1p5b_teacher_forced_decode_matches_the_oracle(rig15, mode):
    got, want = forced_decode(rig15, TOKENS, mode)
    check(rig15, f"1.5B decode mode {mode}", got, want)
    # recurrent state after 16 tokens (f32 [L, S+2, D]); WKV rows are sums of O(1) outer products
    hs, os_ = rig15.rt.state_back(0), rig15.oracle.state
    d = np.abs(hs - os_)
    record(f"1.5B state mode {mode}", {"max": float(d.max()), "mean": float(d.mean()), "ref_absmax": float(np.abs(os_).max())})
    assert d.mean() <= 2e-3 and d.max() <= 2e-2 * max(1.0, float(np.abs(os_).max())), (d.max(), d.mean())


@pytest.mark.parametrize("mode", [1, 0])
def test_1p5b_prefill_chunk_matches_the_oracle(rig15, mode):
    """One 96-token chunk of one sequence (MFMA tile GEMMs, chunk WKV kernel, merged launches in mode 1) vs the oracle fed the
    same 96 tokens sequentially; then 4 teacher-forced decode steps on the carried state."""
    V = rig15.V
    prompt = [(31 * i + 5) % (V - 1) for i in range(96)]
    rig15.zero()
    got = rig15.rt.infer(wrk.RnnInput([prompt], 128), mode=mode)[0]
    for t in prompt:
        want = rig15.oracle.decode(t).copy()
    assert got.shape == (1, V)
    rows_g, rows_w = [got[0]], [want]
    for t in TOKENS[:4]:
        rows_g.append(rig15.rt.infer_raw([t], [1 << 24], [0], mode=mode)[0].copy())
        rows_w.append(rig15.oracle.decode(int(t)).copy())
    check(rig15, f"1.5B prefill96+4 mode {mode}", np.stack(rows_g), np.stack(rows_w))


def test_2p9b_shape_teacher_forced_decode_matches_the_oracle():
    """cfg 3's architecture (32 layers, D = 2560: K is not a multiple of 2048, rows of 10 super-blocks) for a few tokens."""
    rig = Rig("2.9B", 11)
    try:
        for mode in (1, 0):
            got, want = forced_decode(rig, TOKENS[:4], mode)
            check(rig, f"2.9B decode mode {mode}", got, want)
    finally:
        rig.close()
