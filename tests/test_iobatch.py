"""Corpus pipeline (magphase_amd/iobatch.py): ordering, overlap bookkeeping and error propagation of the three stages."""
import threading
import time

import pytest

from magphase_amd import iobatch


def test_batches_split():
    assert iobatch.batches(range(7), 3) == [[0, 1, 2], [3, 4, 5], [6]]
    assert iobatch.batches([], 4) == []


def test_pipeline_preserves_order_and_runs_stages_in_their_own_threads():
    seen = {"load": set(), "store": set()}
    stored = []

    def load(w):
        seen["load"].add(threading.get_ident())
        time.sleep(0.002)
        return w * 10

    def compute(x):
        assert threading.get_ident() == main
        return x + 1

    def store(y):
        seen["store"].add(threading.get_ident())
        stored.append(y)

    main = threading.get_ident()
    n = iobatch.pipeline(list(range(12)), load, compute, store)
    assert n == 12 and stored == [w * 10 + 1 for w in range(12)]
    assert main not in seen["load"] and main not in seen["store"] and seen["load"] != seen["store"]


@pytest.mark.parametrize("stage", ["load", "compute", "store"])
def test_pipeline_reraises_stage_errors(stage):
    def boom(tag):
        def f(x):
            if tag == stage and x >= 3:
                raise RuntimeError("failed in " + tag)
            return x
        return f

    with pytest.raises(RuntimeError, match="failed in " + stage):
        iobatch.pipeline(list(range(8)), boom("load"), boom("compute"), boom("store"))


def test_reader_runs_ahead_of_compute():
    """The reader has the next item ready while compute works on the current one (bounded look-ahead)."""
    loaded_at, computed_at = {}, {}

    def load(w):
        loaded_at[w] = time.time()
        return w

    def compute(w):
        time.sleep(0.02)
        computed_at[w] = time.time()
        return w

    iobatch.pipeline(list(range(5)), load, compute, lambda r: None, depth=2)
    assert all(loaded_at[w + 1] < computed_at[w] for w in range(4))


def test_keyboard_interrupt_in_compute_does_not_hang():
    """Ctrl-C in the calling thread: the stages are told to stop and never block on a full queue."""
    def load(w):
        return w

    def compute(w):
        if w == 2:
            raise KeyboardInterrupt
        return w

    t0 = time.time()
    with pytest.raises(KeyboardInterrupt):
        iobatch.pipeline(list(range(50)), load, compute, lambda r: time.sleep(0.01), depth=2)
    assert time.time() - t0 < 5.0


def test_isolate_retries_one_by_one_and_crash_list(tmp_path):
    def fn(batch):
        if any(x == 3 for x in batch):
            raise ValueError("bad item")
        return [x * 2 for x in batch]

    ok, failed = iobatch._isolate([1, 2, 3, 4], fn)
    assert ok == [(0, 2), (1, 4), (3, 8)] and [i for i, _e in failed] == [2]
    ok, failed = iobatch._isolate([1, 2], fn)
    assert ok == [(0, 2), (1, 4)] and failed == []
    rep = iobatch.CorpusReport()
    iobatch._record_failures(rep, str(tmp_path), [("tok_a", "ValueError: x"), ("tok_b", "IOError: y")])
    iobatch._record_failures(rep, str(tmp_path), [])
    path = rep["crash_list"]
    import os
    import socket
    assert os.path.basename(path) == "crash_file_list_%s_%d.scp" % (socket.gethostname(), os.getpid())
    assert open(path).read().split() == ["tok_a", "tok_b"] and len(rep["failed"]) == 2
