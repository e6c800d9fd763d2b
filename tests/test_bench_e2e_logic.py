"""The orchestration of bench.py's end-to-end leg (reader | device workers | tail workers, warm-up through the same threads, watchdog,
error propagation) with stand-ins for the library: no GPU, no index."""
import sys
import os
import time
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


class _Chunk:
    def __init__(self, t1, t2, n_threads):
        self.n_reads = len(t1)
        self.f = types.SimpleNamespace(n_bases=150 * self.n_reads)
        self.tag = t1

    def close(self):
        pass


class _Ctx:
    log = []

    def __init__(self, share=None, fail_at=None, tail_sleep=0.0):
        self.fail_at = share.fail_at if share is not None else fail_at
        self.tail_sleep = share.tail_sleep if share is not None else tail_sleep
        self.cur = None

    def batch_upload_chunk(self, ch):
        self.cur = ch

    def batch_run(self, opt):
        time.sleep(0.002)
        if self.fail_at is not None and self.cur.tag == self.fail_at:
            raise RuntimeError("device stage failed on purpose")

    def batch_finish(self, opt):
        pass

    def batch_download_alnregs(self, out=None):
        if out is None:                                            # (the serial re-run of the last chunk: no pool)
            return np.zeros(1), np.zeros(2, np.int64)
        assert len(out) >= 3 * self.cur.n_reads                    # the pool's page-locked buffer
        return out[:1], np.zeros(2, np.int64)

    def sam(self, ch, opt, so, aln, aln_off, n_before, paired, out=None):
        time.sleep(self.tail_sleep)
        if out is None:                                            # (the serial re-run: not one of the pipeline's chunks)
            return np.full(10 * ch.n_reads, ch.tag[0], np.uint8)
        _Ctx.log.append((ch.tag, n_before))
        out[:10 * ch.n_reads] = ch.tag[0]
        return out[:10 * ch.n_reads]

    def close(self):
        pass


class _Pinned:
    live = 0

    def __init__(self, n, dtype=np.uint8):
        self.a = np.zeros(n, dtype)
        _Pinned.live += 1

    def close(self):
        _Pinned.live -= 1


def _fake_bm2(**kw):
    return types.SimpleNamespace(Context=lambda share=None: _Ctx(share=share), FastqChunk=_Chunk, host_cpus=lambda: 64, Pinned=_Pinned, ALNREG_DT=np.uint8,
                                 default_sam_opt=lambda n_threads=0: types.SimpleNamespace(n_threads=n_threads))


def test_pipeline_counts_only_the_timed_chunks_and_seeds_the_read_numbers():
    _Ctx.log = []
    texts = [(bytes([i]) * (100 + i), None) for i in range(7)]
    r = bench.end_to_end(_Ctx(), _fake_bm2(), texts, None, True, 2, limit_s=30)
    assert r["chunks"] == 7 and r["warmup_chunks"] == 3 and r["device_workers"] == 2 and r["tail_workers"] == 3
    assert r["reads"] == sum(100 + i for i in range(7)) and r["sam_bytes"] == 10 * r["reads"]
    timed = [x for x in _Ctx.log][-7:]                            # the warm-up chunks come first, then every chunk once more
    before = {tag[0]: nb for tag, nb in timed}
    assert before == {i: sum(100 + j for j in range(i)) for i in range(7)}
    assert _Pinned.live == 0                                       # every pool buffer was released
    assert r["chunk_check"] == {"chunk": 6, "bytes": 10 * 106, "equal_to_serial_run": True}


def test_an_error_in_a_stage_is_raised_not_waited_for():
    texts = [(bytes([i]) * 50, None) for i in range(6)]
    t = time.time()
    with pytest.raises(RuntimeError):
        bench.end_to_end(_Ctx(fail_at=bytes([4]) * 50), _fake_bm2(), texts, None, True, 2, limit_s=30)
    assert time.time() - t < 10


def test_the_watchdog_ends_a_stuck_leg():
    texts = [(bytes([i]) * 50, None) for i in range(4)]
    with pytest.raises(TimeoutError):
        bench.end_to_end(_Ctx(tail_sleep=5.0), _fake_bm2(), texts, None, True, 2, limit_s=1.0)
