"""GPU parity of every TensorOp kernel against the oracle (through the C ABI, on real hardware).

Tolerances: every op computes in f32 and stores f16 where the reference does.  An f16 store may
land one f16 ulp away from the oracle's when the f32 value sits next to a rounding boundary, so
f16 outputs are compared with atol = 1 f16 ulp of the value's magnitude plus 1e-6; f32 outputs
(state) with rtol 2e-6.
"""
import numpy as np
import pytest

import wrk
from oracle import rwkv7 as O
from oracle.rnn import stack_cursors

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = wrk.Context(0)
    yield c
    c.close()


def rng(seed):
    return np.random.default_rng(seed)


def h16(a):
    return np.asarray(a, np.float32).astype(np.float16)


def close16(got, want, ulps=1.0):
    got = np.asarray(got, np.float32)
    want = np.asarray(want, np.float32)
    tol = ulps * np.maximum(np.abs(want), 2.0 ** -14) * 2.0 ** -10 + 1e-6
    bad = np.abs(got - want) > tol
    assert not bad.any(), f"{bad.sum()} / {bad.size} off; worst {np.abs(got - want).max():.3e}"


@pytest.mark.parametrize("C,T,B", [(1000, 3, 2), (768, 1, 1), (2048, 5, 1), (64, 2, 3)])
def test_layer_norm(ctx, C, T, B):
    r = rng(C)
    x = h16(10 * (r.random((B, T, C)) - 0.5))
    w, b = h16(r.random(C) - 0.5), h16(r.random(C) - 0.5)
    t = ctx.tensor(x)
    wrk.TensorOp.layer_norm(ctx.buffer(w), ctx.buffer(b), t, 1e-5)
    want = O.r16(O.layer_norm(x.astype(np.float32), w.astype(np.float32), b.astype(np.float32), np.float32(1e-5)))
    close16(t.back().reshape(B, T, C), want)


def test_layer_norm_f32_buffer(ctx):
    r = rng(5)
    x = (10 * (r.random((2, 3, 1000)) - 0.5)).astype(np.float32)
    w, b = h16(r.random(1000) - 0.5), h16(r.random(1000) - 0.5)
    t = ctx.tensor(x)
    wrk.TensorOp.layer_norm(ctx.buffer(w), ctx.buffer(b), t, 1e-5)
    want = O.layer_norm(x, w.astype(np.float32), b.astype(np.float32), np.float32(1e-5))
    # the reference's own bound for this op is 1e-3 (ops.rs:3481)
    np.testing.assert_allclose(t.back().reshape(x.shape), want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("H,T", [(4, 3), (32, 1)])
def test_group_norm_and_l2_norm(ctx, H, T):
    S = 64
    r = rng(H)
    x = h16(4 * (r.random((T, H, S)) - 0.5))
    w, b = h16(1 + 0.1 * r.standard_normal(H * S)), h16(0.1 * r.standard_normal(H * S))
    t = ctx.tensor(x)
    wrk.TensorOp.group_norm(ctx.buffer(w), ctx.buffer(b), t, 64e-5)
    want = O.r16(O.layer_norm(x.astype(np.float32), w.astype(np.float32).reshape(H, S)[None], b.astype(np.float32).reshape(H, S)[None], np.float32(64e-5)))
    close16(t.back().reshape(T, H, S), want)
    t = ctx.tensor(x)
    wrk.TensorOp.l2_norm(t, 1e-12)
    close16(t.back().reshape(T, H, S), O.r16(O.l2_norm(x.astype(np.float32), np.float32(1e-12))))


def test_token_shift_ragged_batches(ctx):
    C, B = 256, 3
    lens = [3, 0, 4]
    T = sum(lens)
    r = rng(1)
    x = h16(r.standard_normal((T, C)))
    mu = h16(r.random(C))
    state = r.standard_normal((B, 66, C)).astype(np.float32)
    cur = np.array(stack_cursors(lens), np.uint32)
    st = ctx.tensor(state)                                  # [C, 66, B]
    out = ctx.zeros([C, T, 1])
    for row in (0, 65):
        wrk.TensorOp.token_shift(ctx.buffer(cur), ctx.tensor(mu, [C, 1, 1]), st.view(None, row), ctx.tensor(x, [C, T, 1]), out, True)
        prev = np.empty((T, C), np.float32)
        prev[1:] = x[:-1]
        prev[0] = state[0, row]
        prev[3] = state[2, row]
        want = O.r16(O.mix(x.astype(np.float32), prev, mu.astype(np.float32)[None]))
        close16(out.back().reshape(T, C), want)
    # not reversed: mix(prev, x, mu)
    wrk.TensorOp.token_shift(ctx.buffer(cur), ctx.tensor(mu, [C, 1, 1]), st.view(None, 0), ctx.tensor(x, [C, T, 1]), out, False)
    prev[0], prev[3] = state[0, 0], state[2, 0]
    close16(out.back().reshape(T, C), O.r16(O.mix(prev, x.astype(np.float32), mu.astype(np.float32)[None])))


def test_binary_lerp_control_affine_blit(ctx):
    C, T = 512, 4
    r = rng(2)
    a, b = h16(r.standard_normal((T, C))), h16(r.standard_normal((T, C)))
    v = h16(r.standard_normal(C))
    af, bf = a.astype(np.float32), b.astype(np.float32)
    # add with broadcast + sigmoid on the output (v7.rs:867-873)
    out = ctx.tensor(b, [C, T, 1])
    wrk.TensorOp.add_activate(ctx.tensor(v, [C, 1, 1]), out, "none", "none", "sigmoid")
    close16(out.back().reshape(T, C), O.r16(O.sigmoid(v.astype(np.float32)[None] + bf)), 2)
    out = ctx.tensor(b, [C, T, 1])
    wrk.TensorOp.mul(ctx.tensor(a, [C, T, 1]), out)
    close16(out.back().reshape(T, C), O.r16(af * bf))
    # lerp reversed: y <- mix(y, x, f)
    f = h16(r.random((T, C)))
    y = ctx.tensor(b, [C, T, 1])
    wrk.TensorOp.lerp(ctx.tensor(a, [C, T, 1]), y, ctx.tensor(f, [C, T, 1]), True)
    close16(y.back().reshape(T, C), O.r16(O.mix(bf, af, f.astype(np.float32))))
    # control_k: k * (1 + (a - 1) * p)
    k = ctx.tensor(b, [C, T, 1])
    wrk.TensorOp.control_k_v7(ctx.buffer(v), ctx.tensor(a, [C, T, 1]), k)
    close16(k.back().reshape(T, C), O.r16(bf * (1 + (af - 1) * v.astype(np.float32)[None])))
    # affine + blit (f16 -> f32 slice of a [C, T, 4] tensor)
    x = ctx.tensor(a, [C, T, 1])
    wrk.TensorOp.affine(x, 0.5, 0.0)
    close16(x.back().reshape(T, C), O.r16(0.5 * af))
    n = ctx.zeros([C, T, 4], np.float32)
    wrk.TensorOp.blit(ctx.tensor(a, [C, T, 1]), n.view(None, None, 2))
    got = n.back().reshape(4, T, C)
    assert np.array_equal(got[2], af) and not got[[0, 1, 3]].any()


def test_shape_errors_are_reported_not_launched(ctx):
    a = ctx.zeros([64, 2, 1])
    b = ctx.zeros([32, 2, 1])
    with pytest.raises(wrk.WrkError):
        wrk.TensorOp.blit(a, b)
    with pytest.raises(wrk.WrkError):
        wrk.TensorOp.mul(a, b)
    bad = wrk.Tensor(ctx, a.buf, wrk.F16, [64, 4, 1, 1])     # view larger than its buffer
    with pytest.raises(wrk.WrkError):
        wrk.TensorOp.affine(bad, 1.0, 0.0)


def _wkv_reference(state, r, w, k, v, a, kk, x_ln, lens, H):
    """oracle restatement of time_mix_v7.wgsl:143-221 for stacked tokens (float32)."""
    S = 64
    T, D = r.shape
    cur = stack_cursors(lens)
    y = np.empty((T, D), np.float32)
    ww = np.exp(O.W_SCALE * O.sigmoid(w), dtype=np.float32)
    aa, bb = -kk, kk * a
    st = state.copy()
    for t in range(T):
        b, start, n = cur[t] & 0xFF, (cur[t] >> 8) & 0xFFFF, cur[t] >> 24
        if t - start + 1 == n:
            st[b, 0] = x_ln[start + n - 1]
        Sm = st[b, 1:S + 1].reshape(S, H, S).transpose(1, 0, 2)
        sa = np.einsum("hj,hji->hi", aa[t].reshape(H, S), Sm).astype(np.float32)
        Sn = (Sm * ww[t].reshape(H, S)[:, :, None] + k[t].reshape(H, S)[:, :, None] * v[t].reshape(H, S)[:, None, :]
              + sa[:, None, :] * bb[t].reshape(H, S)[:, :, None]).astype(np.float32)
        y[t] = np.einsum("hj,hji->hi", r[t].reshape(H, S), Sn).reshape(D)
        st[b, 1:S + 1] = Sn.transpose(1, 0, 2).reshape(S, D)
    return y, st


@pytest.mark.parametrize("lens,H", [([1], 4), ([3, 0, 5], 4), ([1, 1, 1, 1], 32)])
def test_time_mix_v7_and_time_first(ctx, lens, H):
    S, B = 64, len(lens)
    D, T = H * S, sum(lens)
    r_ = rng(H + T)
    g = lambda s=1.0: h16(s * r_.standard_normal((T, D)))
    r, w, k, v, xln = g(), g(), g(0.5), g(), g()
    a = h16(r_.random((T, D)))
    kk = h16(O.l2_norm(r_.standard_normal((T, H, S)).astype(np.float32), np.float32(1e-12)).reshape(T, D))
    state = (0.3 * r_.standard_normal((B, S + 2, D))).astype(np.float32)
    cur = ctx.buffer(np.array(stack_cursors(lens), np.uint32))
    st = ctx.tensor(state)
    n = ctx.tensor(np.stack([k, v, a, kk]), [S, H, T, 4])
    x = ctx.tensor(xln, [S, H, T])
    rt, wt = ctx.tensor(r, [S, H, T]), ctx.tensor(w, [S, H, T])
    wrk.TensorOp.time_mix_v7(cur, st.view(None, (0, S + 1)), rt, wt, n, x)
    f = lambda z: z.astype(np.float32)
    want_y, want_st = _wkv_reference(state, f(r), f(w), f(k), f(v), f(a), f(kk), f(xln), lens, H)
    got_st = st.back().reshape(B, S + 2, D)
    np.testing.assert_allclose(got_st, want_st, rtol=3e-5, atol=3e-5)
    assert np.array_equal(got_st[:, S + 1], state[:, S + 1])          # ffn row untouched
    close16(x.back().reshape(T, D), O.r16(want_y), 2)
    # time_first on top: x += (sum_j u k r) v
    u = h16(0.3 * r_.standard_normal(D))
    y16 = x.back().reshape(T, D).astype(np.float32)
    wrk.TensorOp.time_first_v7(ctx.buffer(u), rt, n, x)
    xx = (f(u)[None] * f(k) * f(r)).reshape(T, H, S).sum(-1)
    want = O.r16(y16 + (xx[:, :, None] * f(v).reshape(T, H, S)).reshape(T, D))
    close16(x.back().reshape(T, D), want, 2)


@pytest.mark.parametrize("lens,H", [([7, 1, 13, 4], 2), ([70, 3, 66], 1), ([0, 5, 0, 2], 3), ([200] * 6, 1)])
def test_time_mix_v7_chunk_kernels_agree(ctx, lens, H, monkeypatch):
    """The chunk kernels of the dense f16 layout -- four or eight threads per state column (few sequences) and one wave per head (many) --
    against the oracle and against each other.  Tails of the 4-token prefetch ring, empty batches, more
    than 64 and more than 1024 stacked tokens (the in-kernel search for the s-th sequence)."""
    S, B = 64, len(lens)
    D, T = H * S, sum(lens)
    r_ = rng(17 * H + T)
    g = lambda s=1.0: h16(s * r_.standard_normal((T, D)))
    r, w, k, v, xln = g(), g(), g(0.5), g(), g()
    a = h16(r_.random((T, D)))
    kk = h16(O.l2_norm(r_.standard_normal((T, H, S)).astype(np.float32), np.float32(1e-12)).reshape(T, D))
    state = (0.3 * r_.standard_normal((B, S + 2, D))).astype(np.float32)
    f = lambda z: z.astype(np.float32)
    want_y, want_st = _wkv_reference(state, f(r), f(w), f(k), f(v), f(a), f(kk), f(xln), lens, H)
    got = {}
    for name, wave, octs in (("quad", "0", "0"), ("wave", "1", "0"), ("oct", "0", "1")):
        monkeypatch.setenv("WRK_WKV_WAVE", wave)
        monkeypatch.setenv("WRK_WKV_OCT", octs)
        cur = ctx.buffer(np.array(stack_cursors(lens), np.uint32))
        st = ctx.tensor(state)
        n = ctx.tensor(np.stack([k, v, a, kk]), [S, H, T, 4])
        x = ctx.tensor(xln, [S, H, T])
        wrk.TensorOp.time_mix_v7(cur, st.view(None, (0, S + 1)), ctx.tensor(r, [S, H, T]), ctx.tensor(w, [S, H, T]), n, x)
        got[name] = (st.back().reshape(B, S + 2, D), x.back().reshape(T, D))
        # decay compounds over up to 200 steps: compare where the oracle's own f32 rounding leaves room
        np.testing.assert_allclose(got[name][0], want_st, rtol=2e-4, atol=2e-4)
        close16(got[name][1], O.r16(want_y), 3)
    # four waves per head and one wave per head: the same summation orders, bit for bit; eight threads per column (few sequences) splits each
    # chain of four in two: equal within the oracle bound only
    assert np.array_equal(got["quad"][0], got["wave"][0]) and np.array_equal(got["quad"][1].view(np.uint16), got["wave"][1].view(np.uint16))


def test_channel_mix_v7(ctx):
    C, lens = 256, [2, 3]
    T, B = sum(lens), 2
    r_ = rng(9)
    v, x = h16(r_.standard_normal((T, C))), h16(r_.standard_normal((T, C)))
    state = r_.standard_normal((B, 66, C)).astype(np.float32)
    st, xt = ctx.tensor(state), ctx.tensor(x, [C, T, 1])
    wrk.TensorOp.channel_mix_v7(ctx.buffer(np.array(stack_cursors(lens), np.uint32)), st.view(None, 65), ctx.tensor(v, [C, T, 1]), xt)
    got = st.back().reshape(B, 66, C)
    assert np.array_equal(got[0, 65], x[1].astype(np.float32)) and np.array_equal(got[1, 65], x[4].astype(np.float32))
    assert np.array_equal(got[:, :65], state[:, :65])
    assert np.array_equal(xt.back().reshape(T, C), v)


def test_softmax(ctx):
    x = rng(3).standard_normal((2, 1000)).astype(np.float32) * 4
    t = ctx.tensor(x, [1000, 2, 1])
    wrk.TensorOp.softmax(t)
    e = np.exp(x - x.max(1, keepdims=True))
    np.testing.assert_allclose(t.back().reshape(2, 1000), e / e.sum(1, keepdims=True), rtol=1e-5, atol=1e-8)
