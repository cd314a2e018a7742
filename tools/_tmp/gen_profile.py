import cProfile, pstats, sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
from magphase_amd import engine as em
eng = em.get_engine()
utts = bench.make_batch(0)[:32]
st = bench._lowdim_state(em, eng, utts)
aplan = st["aplan"]
res = [t.cpu().numpy().astype(np.float32) for t in st["out"]]
from scipy import signal
sutts = []
for u in range(len(utts)):
    a, b = int(aplan.out_off[u]), int(aplan.out_off[u + 1])
    v_f0 = aplan.f0_out[u]
    with np.errstate(divide="ignore"):
        v_lf0 = np.log((v_f0 > 0).astype(float) * signal.medfilt(v_f0))
    v_lf0[np.isinf(v_lf0) | np.isnan(v_lf0)] = -1.0e10
    sutts.append((res[0][a:b], res[1][a:b], res[2][a:b], v_lf0))
def one(mode):
    p = em.CompressedSynthesisPlan(eng, sutts, bench.FS, b_const_rate=True, post_filter=True, noise_mode=mode)
    pcm = p.run()
    torch.cuda.synchronize()
    return p
for mode in ("reference", "device"):
    for _ in range(3): one(mode)
    ts = []
    for _ in range(8):
        t = time.perf_counter(); one(mode); ts.append(time.perf_counter() - t)
    print(mode, "plan+run ms:", ["%.2f" % (x * 1e3) for x in ts])
pr = cProfile.Profile(); pr.enable()
for _ in range(10): one("reference")
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
