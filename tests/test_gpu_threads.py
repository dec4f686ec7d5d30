"""The boundary honours the reference's threading.

`TokioRuntime` encodes speculative jobs on `spawn_blocking` workers (runtime/mod.rs:139-167, spawn at :165) while the
runtime task submits cached jobs, and a dedicated thread per context blocks in read-backs (context.rs:148-162).
Here: two threads encode different programs at the same time (their `wrk_op_*` calls interleaved on purpose), a third
keeps launching an already encoded program and reading its output, a fourth uploads and allocates.  Every result
must equal the serial run bit for bit, and nothing an encoder records may execute before its program is launched.
"""
import threading

import numpy as np
import pytest

import wrk

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = wrk.Context(0)
    yield c
    c.close()


def h16(a):
    return np.asarray(a, np.float32).astype(np.float16)


class Job:
    """A small op list on private buffers: x <- LN(x); y = relu(W x)^2; x2 <- x2 + y (n_rep times)."""

    def __init__(self, ctx, seed, k, m, t, kind, n_rep):
        r = np.random.default_rng(seed)
        self.ctx, self.n_rep, self.t = ctx, n_rep, t
        self.x0 = h16(r.standard_normal((t, k)))
        self.acc0 = h16(r.standard_normal((t, m)))
        self.w, self.b = ctx.buffer(h16(1 + 0.1 * r.standard_normal(k))), ctx.buffer(h16(0.1 * r.standard_normal(k)))
        if kind == "F16":
            self.mat = wrk.Matrix(ctx, "F16", k, m, h16(r.standard_normal((m, k)) / np.sqrt(k)))
        else:
            from oracle import quantize
            self.mat = wrk.Matrix(ctx, kind, k, m, quantize.QUANTIZE[kind]((r.standard_normal(m * k) / np.sqrt(k)).astype(np.float32)))
        self.x = ctx.tensor(self.x0)
        self.y = ctx.zeros([m, t], np.float16)
        self.acc = ctx.tensor(self.acc0)

    def reset(self):
        self.x.buf.write(self.x0)
        self.acc.buf.write(self.acc0)
        self.y.buf.write(np.zeros(self.y.shape[0] * self.t, np.float16))

    def ops(self, pause=None):
        for i in range(self.n_rep):
            wrk.TensorOp.layer_norm(self.w, self.b, self.x, 1e-5)
            if pause:
                pause()
            self.mat.matmul_op(self.x, self.y, act="squared_relu", turbo=self.t >= 4)
            if pause:
                pause()
            wrk.TensorOp.add(self.y, self.acc)

    def result(self):
        return self.x.back().copy(), self.acc.back().copy()


def test_concurrent_encoders_launcher_and_uploader(ctx):
    ja = Job(ctx, 1, 512, 384, 1, "Q4_K", 2)       # same op count as jb: the encoders meet after every op
    jb = Job(ctx, 2, 768, 256, 8, "F16", 2)
    jc = Job(ctx, 3, 256, 512, 2, "Q6_K", 1)
    # serial references (eager)
    want = {}
    for name, j in (("a", ja), ("b", jb), ("c", jc)):
        j.reset()
        j.ops()
        ctx.sync()
        want[name] = j.result()
        j.reset()
    ctx.sync()
    pc = ctx.encode(jc.ops)                     # the "cached job" the runtime task keeps submitting

    meet = threading.Barrier(2, timeout=60)
    progs, errors, c_runs = {}, [], []
    stop = threading.Event()

    def encoder(name, job):
        try:
            progs[name] = ctx.encode(lambda: job.ops(pause=meet.wait))      # the two encoders alternate op by op
        except Exception as e:      # noqa: BLE001
            errors.append((name, e))
            meet.abort()

    def launcher():
        try:
            while not stop.is_set() or len(c_runs) < 3:
                jc.reset()
                pc.launch()
                c_runs.append(jc.result())          # blocking read-back while the others are mid-capture
        except Exception as e:      # noqa: BLE001
            errors.append(("launcher", e))

    def uploader():
        try:
            r = np.random.default_rng(9)
            while not stop.is_set():
                a = r.standard_normal(4096).astype(np.float32)
                buf = ctx.buffer(a)                 # hipMalloc + H2D + stream sync on the submission stream
                assert np.array_equal(buf.read(np.float32, 4096), a)
        except Exception as e:      # noqa: BLE001
            errors.append(("uploader", e))

    ts = [threading.Thread(target=encoder, args=("a", ja)), threading.Thread(target=encoder, args=("b", jb)),
          threading.Thread(target=launcher), threading.Thread(target=uploader)]
    for t in ts:
        t.start()
    ts[0].join(120)
    ts[1].join(120)
    stop.set()
    ts[2].join(120)
    ts[3].join(120)
    assert not errors, errors
    assert not any(t.is_alive() for t in ts)
    # encoding executed nothing: the buffers still hold their initial values
    ctx.sync()
    for j in (ja, jb):
        x, acc = j.result()
        assert np.array_equal(x.reshape(j.x0.shape), j.x0) and np.array_equal(acc.reshape(j.acc0.shape), j.acc0)
    # every concurrent launch of the cached program equals the serial run
    assert len(c_runs) >= 3
    for x, acc in c_runs:
        assert np.array_equal(x, want["c"][0]) and np.array_equal(acc, want["c"][1])
    # the concurrently encoded programs replay to the serial results, twice (a program is reusable like a CommandBuffer is not:
    # the reference re-encodes; here the same graph is launched again)
    for _ in range(2):
        for name, j in (("a", ja), ("b", jb)):
            j.reset()
            progs[name].launch()
            x, acc = j.result()
            assert np.array_equal(x, want[name][0]) and np.array_equal(acc, want[name][1]), name


def test_capture_is_per_thread(ctx):
    """A second begin on the SAME thread is an error; an end without a begin is an error; another thread may begin meanwhile."""
    assert wrk.hip.wrk_capture_begin(ctx.h) == 0
    assert wrk.hip.wrk_capture_begin(ctx.h) == wrk.E_ARG
    out = {}

    def other():
        out["rc_end_without_begin"] = wrk.hip.wrk_capture_end(ctx.h, wrk.C.byref(wrk._P()))
        out["prog"] = ctx.encode(lambda: None)                  # an empty program from another thread, while ours is open

    t = threading.Thread(target=other)
    t.start()
    t.join(60)
    assert out["rc_end_without_begin"] == wrk.E_ARG
    h = wrk._P()
    assert wrk.hip.wrk_capture_end(ctx.h, wrk.C.byref(h)) == 0
    wrk.Program(ctx, h).launch()
    out["prog"].launch()
    ctx.sync()


def test_model_jobs_from_two_threads(ctx):
    """Two threads drive `runtime.infer` on two runtimes of one context (each call captures its job graph on a private
    stream and launches it on the submission stream): results equal the single-threaded run."""
    from oracle import synth
    data = synth.make_v7_gguf(synth.CONFIGS["tiny"], 42)
    V = synth.CONFIGS["tiny"].num_vocab
    prompts = [synth.tokens(11, "p0", 40, V), synth.tokens(12, "p1", 40, V)]

    def run(rt, prompt, out, i):
        inp = wrk.RnnInput([prompt], 32)
        a = rt.infer(inp, mode=1)[0]
        b = rt.infer(inp, mode=1)[0]
        out[i] = (a, b, rt.state_back(0))

    serial, threaded = {}, {}
    for i in range(2):
        rt = wrk.Runtime(ctx, wrk.GgufReader(data), num_batch=1)
        run(rt, prompts[i], serial, i)
        rt.close()
    rts = [wrk.Runtime(ctx, wrk.GgufReader(data), num_batch=1) for _ in range(2)]
    ts = [threading.Thread(target=run, args=(rts[i], prompts[i], threaded, i)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(120)
    for i in range(2):
        for got, want in zip(threaded[i], serial[i]):
            assert np.array_equal(got, want)
    for rt in rts:
        rt.close()
