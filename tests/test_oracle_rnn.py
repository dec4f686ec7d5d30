"""Chunk scheduler oracle vs the reference's own known-answer tests.

Expected values are the data of /root/reference/src/runtime/infer/rnn.rs:363-569
(test_run_iter, test_advance, test_redirect); they pin oracle/rnn.py.
"""
from oracle.rnn import FULL, LAST, RnnInput, RnnInputBatch, pack_cursor, stack_cursors


def _run(lens_opts, chunk):
    inp = RnnInput([RnnInputBatch([i] * n, o) for i, (n, o) in enumerate(lens_opts)], chunk)
    return inp


def _info(it):
    return [(b.len, b.option) for b in next(it)]


def test_run_iter():
    run = _run([(139, LAST), (1, LAST), (0, FULL), (65, FULL)], 128)
    it = run.iter()
    assert _info(it) == [(65, None), (1, LAST), (0, FULL), (62, FULL)]
    assert _info(it) == [(60, None), (1, LAST), (0, FULL), (3, FULL)]
    assert _info(it) == [(14, LAST), (1, LAST), (0, FULL), (1, FULL)]
    assert _info(it) == [(1, LAST), (1, LAST), (0, FULL), (1, FULL)]
    assert _info(it) == [(1, LAST), (1, LAST), (0, FULL), (1, FULL)]


def test_advance():
    run = _run([(139, LAST), (1, LAST), (0, FULL), (65, FULL)], 128)
    run.step()
    assert _info(run.iter()) == [(61, None), (0, LAST), (0, FULL), (3, FULL)]
    run = _run([(61, LAST), (1, LAST), (0, FULL), (3, FULL)], 128)
    assert _info(run.iter()) == [(60, None), (1, LAST), (0, FULL), (3, FULL)]


def test_redirect():
    run = _run([(61, LAST), (0, LAST), (0, FULL), (3, FULL)], 128)
    red = next(run.iter()).redirect()
    assert red.headers == [60, 61, 62, 63]
    assert red.inputs == [(0, 61), (61, 61), (61, 61), (61, 64)]
    assert red.outputs == [(0, 1), (1, 1), (1, 1), (1, 4)]

    run = _run([(11, LAST), (8, LAST), (9, LAST), (4, LAST)] * 2, 32)
    red = next(run.iter()).redirect()
    assert red.headers == [15, 31]
    assert red.inputs == [(0, 4), (4, 8), (8, 12), (12, 16), (16, 20), (20, 24), (24, 28), (28, 32)]
    assert red.outputs == [(0, 0), (0, 0), (0, 0), (0, 1), (1, 1), (1, 1), (1, 1), (1, 2)]


def test_chunk_size_rounding():
    # RnnInput::new (rnn.rs:204-212): max(32) then next multiple of 32
    assert RnnInput([], 1).token_chunk_size == 32
    assert RnnInput([], 33).token_chunk_size == 64
    assert RnnInput([], 128).token_chunk_size == 128


def test_header_runs():
    red = next(_run([(61, LAST), (0, LAST), (0, FULL), (3, FULL)], 128).iter()).redirect()
    assert red.header_runs(64) == [(60, 63, 0, 4)]
    red = next(_run([(11, LAST), (8, LAST), (9, LAST), (4, LAST)] * 2, 32).iter()).redirect()
    assert red.header_runs(32) == [(15, 15, 0, 1), (31, 31, 1, 2)]
    assert red.header_runs(2) is None          # num_token == num_header -> identity


def test_cursors():
    # Cursor::pack (tensor/mod.rs:53-60) + into_cursors (:70-84)
    assert pack_cursor(3, 0x1234, 7) == 3 | (0x1234 << 8) | (7 << 24)
    assert stack_cursors([2, 0, 3]) == [pack_cursor(0, 0, 2)] * 2 + [pack_cursor(2, 2, 3)] * 3
