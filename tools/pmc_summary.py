"""
Summarises the rocprofv3 --pmc passes of tools/profile_round.sh:
    python tools/pmc_summary.py gpurun_out/prof_<tag> [--install]
writes <dir>/pmc_summary.json (per-kernel means of every counter) and <dir>/traffic.json (HBM bytes per launch =
(2 x FETCH_SIZE + WRITE_SIZE) x 1024: both counters are in KiB, and gfx950's FETCH_SIZE reports half of a coalesced
stream -- MI355X_MICROARCH.md, HBM section), stamped with the sha1 of the kernel sources it was measured on.
--install copies the two into profiles/ (traffic.json is what bench.py reads for roofline.traffic).
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
d = sys.argv[1].rstrip("/")
tag = os.path.basename(d).replace("prof_", "")

# one lowdim step (bench.py measure_lowdim) launches each of these once
LOWDIM = ("k_analysis_f64", "k_mel_warp_mfma", "k_warp_phase_rows", "k_post_filter", "k_mel_unwarp_tiled",
          "k_mel_unwarp_mfma", "k_noise_stats", "k_noise_gains", "k_synth_comp_pair")


def short(k):
    return k.split("(")[0].split("::")[-1].split("<")[0].replace("void ", "").strip()


out, launches = {}, {}
for f in sorted(glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True)):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if not k.startswith("k_"):
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in agg.items():
        out.setdefault(k, {}).update({c: sum(v) / len(v) for c, v in cs.items()})
        launches[k] = max(launches.get(k, 0), max(len(v) for v in cs.values()))
if not out:   # no counter files here (tools/profile_round.sh deletes them after summarising on the GPU box): install-only
    # mode -- work from the summary written there instead of overwriting it with an empty one
    prev = os.path.join(d, "pmc_summary.json")
    if not os.path.exists(prev) or not json.load(open(prev)):
        sys.exit("pmc_summary: no counter_collection.csv under %s/pmc_* and no earlier summary" % d)
    out = json.load(open(prev))
    launches = {k: int(v.get("_launches", 1)) for k, v in out.items()}
else:
    for k in out:
        out[k]["_launches"] = launches[k]
    json.dump(out, open(os.path.join(d, "pmc_summary.json"), "w"), indent=1)

import bench  # noqa: E402

tr = {"csrc_sha1": bench._kernel_source_hash(), "source": "profiles/%s_pmc_summary.json" % tag,
      "note": "hbm_bytes_per_launch = (2*FETCH_SIZE + WRITE_SIZE) KiB -> bytes, mean over the profiled launches"}
for k, v in out.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
        tr[k] = {"fetch_size_kib": v["FETCH_SIZE"], "write_size_kib": v["WRITE_SIZE"], "launches": launches[k],
                 "hbm_bytes_per_launch": (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0}
# the lowdim step's fix-up launch is not separable from the lossless one by name: both are counted in k_ola_fixup's mean
ld = [k for k in LOWDIM if k in tr]
if ld:
    # launches per step: a kernel launched twice per step (the unwarp: magnitude job, then the two phase jobs) has twice
    # the launches of k_noise_stats (once per step); the mean per launch is multiplied back
    per_step = {k: max(1, round(launches[k] / launches["k_noise_stats"])) for k in ld}
    tr["lowdim_step"] = {"kernels": per_step,
                         "hbm_bytes_per_launch": sum(per_step[k] * tr[k]["hbm_bytes_per_launch"] for k in ld),
                         "note": "sum over the kernels of one configs[2] step, launches per step as listed "
                                 "(k_ola_fixup excluded: < 1 %)"}
json.dump(tr, open(os.path.join(d, "traffic.json"), "w"), indent=1)
print(json.dumps({k: (round(v["hbm_bytes_per_launch"] / 1e6, 1) if isinstance(v, dict) else v) for k, v in tr.items()}))
if "--install" in sys.argv:
    shutil.copy(os.path.join(d, "traffic.json"), os.path.join(ROOT, "profiles", "traffic.json"))
    shutil.copy(os.path.join(d, "pmc_summary.json"), os.path.join(ROOT, "profiles", "%s_pmc_summary.json" % tag))
    for n in ("kernel_stats.csv", "bench.json", "bench_profiled.json"):
        if os.path.exists(os.path.join(d, n)):
            shutil.copy(os.path.join(d, n), os.path.join(ROOT, "profiles", "%s_%s" % (tag, n)))
