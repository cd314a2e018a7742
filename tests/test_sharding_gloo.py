"""
Multi-process (world_size 2, gloo, CPU) test of the utterance-sharding harness: the N > 1 path has no data-path
collective; ranks process disjoint utterances and only gather scalars.  The per-shard worker here is the CPU oracle
(tests may use it as a stand-in for the GPU engine, which has the same per-utterance interface).
"""
import os
import subprocess
import sys

import numpy as np

from magphase_amd import sharding

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys
sys.path.insert(0, %(root)r)
import numpy as np
from magphase_amd import sharding, synthetic as syn
from oracle import magphase_oracle as orc
dist = sharding.init_process_group("gloo")
utts = [syn.make_utterance(300 + u, dur_s=0.25 + 0.05 * (u %% 4), fs=16000) for u in range(7)]
costs = [len(p[1]) for p in utts]
def work(idx):
    frames, checksum = 0, 0.0
    for i in idx:
        pcm, pm, voi = utts[i]
        o = orc.analysis_lossless_from_epochs(syn.pcm_to_float(pcm), 16000, pm, voi)
        y = orc.synthesis_from_lossless(o[0], o[1], o[2], o[3], 16000)
        frames += len(o[5]); checksum += float(np.sum(np.abs(y)))
    return {"rank": sharding.dist_env()[0], "idx": [int(i) for i in idx], "frames": frames, "checksum": checksum}
res = sharding.run_sharded(work, costs)
dist.barrier()
if sharding.dist_env()[0] == 0:
    print("RESULT " + json.dumps(res))
dist.destroy_process_group()
'''


def test_shard_by_cost_partitions_and_balances():
    rng = np.random.RandomState(0)
    costs = rng.randint(50, 1000, size=101)
    for w in (1, 2, 3, 8):
        shards = sharding.shard_by_cost(costs, w)
        allidx = np.sort(np.concatenate(shards))
        assert np.array_equal(allidx, np.arange(costs.size))
        loads = [costs[s].sum() for s in shards]
        assert max(loads) - min(loads) <= costs.max()


def test_two_rank_gloo_run_matches_single_process(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    import socket
    with socket.socket() as sk:   # a free rendezvous port (a fixed one can be taken by a parallel run)
        sk.bind(("127.0.0.1", 0))
        port = str(sk.getsockname()[1])
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port, OMP_NUM_THREADS="1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                          "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                         capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    import json
    line = [l for l in out.stdout.splitlines() if l.startswith("RESULT ")][0]
    res = json.loads(line[len("RESULT "):])
    assert sorted(r["rank"] for r in res) == [0, 1]
    idx = sorted(i for r in res for i in r["idx"])
    assert idx == list(range(7))
    # single-process reference (same work, no sharding)
    from magphase_amd import synthetic as syn
    from oracle import magphase_oracle as orc
    frames = 0
    checksum = 0.0
    for u in range(7):
        pcm, pm, voi = syn.make_utterance(300 + u, dur_s=0.25 + 0.05 * (u % 4), fs=16000)
        o = orc.analysis_lossless_from_epochs(syn.pcm_to_float(pcm), 16000, pm, voi)
        y = orc.synthesis_from_lossless(o[0], o[1], o[2], o[3], 16000)
        frames += len(o[5])
        checksum += float(np.sum(np.abs(y)))
    assert sum(r["frames"] for r in res) == frames
    assert abs(sum(r["checksum"] for r in res) - checksum) < 1e-9 * checksum


def test_corpus_spec_shards_cover_the_10k_corpus_evenly():
    """bench.py --workload corpus (tools/corpus_workload.py): every rank derives the same corpus and the same LPT shards;
    at 10 000 utterances / 8 ranks the assigned cost differs by less than 0.1 % between ranks, mixed rates included."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import corpus_workload as cw

    for mixed in (False, True):
        dur, fs = cw.corpus_spec(10000, mixed)
        dur2, fs2 = cw.corpus_spec(10000, mixed)
        assert np.array_equal(dur, dur2) and np.array_equal(fs, fs2)
        assert 4.9 < dur.mean() < 5.1 and dur.min() >= 2.0 and dur.max() <= 8.0
        assert set(fs.tolist()) == ({48000, 16000} if mixed else {48000})
        cost = cw.utterance_cost(dur, fs)
        shards = sharding.shard_by_cost(cost, 8)
        assert sorted(np.concatenate(shards).tolist()) == list(range(10000))
        loads = np.array([cost[s].sum() for s in shards])
        assert loads.max() / loads.mean() < 1.001


def test_rank_core_sets_partition_numa_nodes():
    """sharding.rank_core_sets: the cores of a GPU's NUMA node split evenly among the ranks on that node, disjoint sets;
    without NUMA information an even split of everything allowed (libutils.py:61-62: one worker per core, nothing shared)."""
    from magphase_amd import sharding as sh
    node_of = {0: 0, 1: 0, 2: 0, 3: 0, 4: 1, 5: 1, 6: 1, 7: 1}
    cpus = {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}
    sets = sh.rank_core_sets(list(range(8)), range(256), node_of_device=node_of.get, cpus_of_node=cpus.get)
    assert all(len(x) == 32 for x in sets) and len(set().union(*map(set, sets))) == 256
    for r, x in enumerate(sets):
        assert set(x) <= set(cpus[node_of[r]])
    # no NUMA information (or ranks sharing one device in a test): an even split of the allowed cores
    sets = sh.rank_core_sets([0] * 8, range(64), node_of_device=lambda d: None, cpus_of_node=lambda n: [])
    assert [len(x) for x in sets] == [8] * 8 and len(set().union(*map(set, sets))) == 64
    # a restricted affinity mask is respected; more ranks than cores on a node: they share them
    sets = sh.rank_core_sets([0, 1], [3, 4, 5, 70], node_of_device=node_of.get, cpus_of_node=cpus.get)
    assert sorted(sets[0] + sets[1]) == [3, 4] or set(sets[0]) | set(sets[1]) <= {3, 4, 5}
    sets = sh.rank_core_sets([0, 1, 2], [3, 4], node_of_device=node_of.get, cpus_of_node=cpus.get)
    assert all(set(x) <= {3, 4} and x for x in sets)
