"""Do two fresh HIP streams (default / high priority) overlap consecutive bench steps?  (GPU box)"""
import sys, time; sys.path.insert(0,'.')
import torch, bench
from magphase_amd.engine import LosslessAnalysisPlan, LosslessSynthesisPlan, get_engine
eng = get_engine(); utts = bench.make_batch(0)
aplan = LosslessAnalysisPlan(eng, utts); splan = LosslessSynthesisPlan(eng, aplan.v_f0, aplan.fs, aplan.fft_len)
H, F = aplan.fft_len // 2 + 1, aplan.total_frames
bufs = [(tuple(eng.empty_feats(F, H) for _ in range(3)), eng.empty((max(splan.strip_floats, 1),)), eng.empty((splan.total_out,))) for _ in range(2)]
def enqueue(stream, k):
    f_, s_, p_ = bufs[k]
    with torch.cuda.stream(stream):
        aplan.run(out=f_); splan.run(f_[0], f_[1], f_[2], strips=s_, out=p_)
def block(ss, steps=200):
    for i in range(64): enqueue(ss[i % len(ss)], i % len(ss))
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(steps): enqueue(ss[i % len(ss)], i % len(ss))
    torch.cuda.synchronize(); return (time.perf_counter() - t) / steps * 1e3
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range", lo, hi)
for rep in range(3):
    s0 = torch.cuda.Stream()
    print("one stream %.4f" % block([s0]))
    for name, mk in (("two default streams", lambda: [torch.cuda.Stream(), torch.cuda.Stream()]),
                     ("normal + high priority", lambda: [torch.cuda.Stream(priority=0), torch.cuda.Stream(priority=-1)]),
                     ("two high priority", lambda: [torch.cuda.Stream(priority=-1), torch.cuda.Stream(priority=-1)])):
        print("  %-24s %s" % (name, " ".join("%.4f" % block(mk()) for _ in range(4))), flush=True)
streams, rep = bench.pick_streams(2, enqueue)
print("pick_streams:", rep, "%.4f" % block(streams))
