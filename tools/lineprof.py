"""Minimal per-line wall-clock profiler for chosen functions (sys.settrace on their frames only):

    import lineprof; lineprof.watch(SomeClass.__init__); ...run...; lineprof.report()
"""
import collections
import linecache
import sys
import time

_codes = {}
_acc = collections.defaultdict(float)
_hits = collections.Counter()
_last = {}


def _local(frame, event, arg):
    now = time.perf_counter()
    key = id(frame)
    if key in _last:
        ln, t0 = _last[key]
        _acc[(frame.f_code, ln)] += now - t0
        _hits[(frame.f_code, ln)] += 1
    if event == "return":
        _last.pop(key, None)
    else:
        _last[key] = (frame.f_lineno, time.perf_counter())
    return _local


def _global(frame, event, arg):
    if event == "call" and frame.f_code in _codes:
        _last[id(frame)] = (frame.f_lineno, time.perf_counter())
        return _local
    return None


def watch(*funcs):
    for f in funcs:
        _codes[f.__code__] = f.__qualname__
    sys.settrace(_global)


def stop():
    sys.settrace(None)


def report(top=25):
    stop()
    for code, name in _codes.items():
        rows = [(t, ln) for (c, ln), t in _acc.items() if c is code]
        tot = sum(t for t, _ in rows)
        print("== %s: %.2f ms in all" % (name, tot * 1e3))
        for t, ln in sorted(rows, reverse=True)[:top]:
            print("  %8.3f ms %6d x  L%-5d %s" % (t * 1e3, _hits[(code, ln)], ln, linecache.getline(code.co_filename, ln).rstrip()[:110]))
