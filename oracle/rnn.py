"""Chunk scheduler (oracle; test infrastructure only).

Restates ``/root/reference/src/runtime/infer/rnn.rs``:
  RnnInput::new / step / chunk ...... :204-253
  RnnIter::next ..................... :280-335
  RnnInfo::redirect ................. :41-81
  RnnRedirect::op (header row runs) . :101-134
and ``Cursor::pack`` / ``into_cursors`` (src/tensor/mod.rs:53-84), ``TensorStack`` (:1185-1233).
Pinned by the reference's own known-answer tests (rnn.rs:363-569); see tests/test_oracle_rnn.py.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Tuple

MIN_TOKEN_CHUNK_SIZE = 32

LAST = "Last"
FULL = "Full"


@dataclass
class RnnInfoBatch:
    len: int = 0
    option: Optional[str] = None


@dataclass
class RnnRedirect:
    headers: List[int] = field(default_factory=list)
    inputs: List[Tuple[int, int]] = field(default_factory=list)
    outputs: List[Tuple[int, int]] = field(default_factory=list)

    def header_runs(self, num_token: int):
        """rnn.rs:101-134: contiguous (first,last,start,end) blits; empty when identity."""
        n = len(self.headers)
        if num_token == 1 or num_token == n:
            return None
        runs, start, end = [], 0, 1
        while end <= n:
            if end == n or self.headers[end - 1] + 1 != self.headers[end]:
                runs.append((self.headers[start], self.headers[end - 1], start, end))
                start = end
            end += 1
        return runs


class RnnInfo(list):
    def num_token(self) -> int:
        return sum(b.len for b in self)

    def num_batch(self) -> int:
        return len(self)

    def redirect(self) -> RnnRedirect:
        headers: List[int] = []
        inputs = [(0, 0)] * len(self)
        outputs = [(0, 0)] * len(self)
        p_in = p_out = 0
        for b, info in enumerate(self):
            n = info.len
            if info.option is None:
                inputs[b] = (p_in, p_in + n)
                outputs[b] = (p_out, p_out)
                p_in += n
            elif info.option == LAST:
                inputs[b] = (p_in, p_in + n)
                if n == 0:
                    outputs[b] = (p_out, p_out)
                else:
                    outputs[b] = (p_out, p_out + 1)
                    headers.append(p_in + n - 1)
                    p_out += 1
                p_in += n
            else:
                inputs[b] = (p_in, p_in + n)
                outputs[b] = (p_out, p_out + n)
                headers.extend(range(p_in, p_in + n))
                p_out += n
                p_in += n
        return RnnRedirect(headers, inputs, outputs)


@dataclass
class RnnInputBatch:
    tokens: List[int]
    option: str = LAST


class RnnIter:
    """rnn.rs:273-335.  Batch state: ('gen',) or ('read', n)."""

    def __init__(self, lens_options, token_chunk_size: int):
        self.batches = [[("read", n), opt] for n, opt in lens_options]
        self.token_chunk_size = token_chunk_size

    def __iter__(self):
        return self

    def __next__(self) -> RnnInfo:
        remains = [1 if st[0] == "gen" else st[1] for st, _ in self.batches]
        num_token = min(sum(remains), self.token_chunk_size)
        if num_token > MIN_TOKEN_CHUNK_SIZE:
            num_token -= num_token % MIN_TOKEN_CHUNK_SIZE
        info = [RnnInfoBatch() for _ in remains]
        while num_token > 0:
            pos = [x for x in remains if x > 0]
            mid0 = min(pos) if pos else 0
            for i in range(len(remains)):
                if remains[i] == 0:
                    continue
                mid = min(mid0, num_token)
                num_token -= mid
                info[i].len += mid
                remains[i] -= mid
        for i, (inf, remain) in enumerate(zip(info, remains)):
            if inf.len > 0:
                self.batches[i][0] = ("gen",) if remain == 0 else ("read", remain)
            opt = self.batches[i][1]
            if opt == LAST:
                inf.option = LAST if remain == 0 else None
            else:
                inf.option = FULL
        return RnnInfo(info)


class RnnInput:
    def __init__(self, batches: List[RnnInputBatch], token_chunk_size: int):
        t = max(token_chunk_size, MIN_TOKEN_CHUNK_SIZE)
        t = -(-t // MIN_TOKEN_CHUNK_SIZE) * MIN_TOKEN_CHUNK_SIZE
        self.batches = batches
        self.token_chunk_size = t

    def iter(self) -> RnnIter:
        return RnnIter([(len(b.tokens), b.option) for b in self.batches], self.token_chunk_size)

    def num_token(self) -> int:
        return sum(len(b.tokens) for b in self.batches)

    def step(self) -> None:
        info = next(self.iter())
        for b, inf in zip(self.batches, info):
            b.tokens = b.tokens[inf.len:]

    def chunk(self) -> List[List[int]]:
        info = next(self.iter())
        return [b.tokens[: inf.len] for b, inf in zip(self.batches, info)]


def pack_cursor(batch: int, token: int, length: int) -> int:
    """tensor/mod.rs:53-60: [batch u8][token u16 le][len u8]."""
    return (batch & 0xFF) | ((token & 0xFFFF) << 8) | ((length & 0xFF) << 24)


def stack_cursors(chunk_lens: List[int]) -> List[int]:
    """TensorStack cursors + into_cursors (tensor/mod.rs:70-84, 1201-1214): one packed cursor per token."""
    out, token = [], 0
    for batch, n in enumerate(chunk_lens):
        if n > 0:
            out.extend([pack_cursor(batch, token, n)] * n)
        token += n
    return out
