#!/bin/bash
# Everything a round's profiles/ entry is made of, in ONE call on the GPU box:   tools/round_evidence.sh r06_v2
#   gpurun_out/prof_<tag>/...               tools/profile_round.sh (bench line, rocprofv3 kernel stats, counter passes)
#   gpurun_out/ev_<tag>/gpu_tests.txt       python -m pytest tests -m gpu  (full session; tolerance report beside it)
#   gpurun_out/ev_<tag>/share_device_n8.json  the N = 8 contract line with eight ranks on device 0 (BENCH_SHARE_DEVICE=1, gloo)
#   gpurun_out/ev_<tag>/corpus1250.json     bench.py --workload corpus --utts 1250
#   gpurun_out/ev_<tag>/corpus_busy.json    tools/corpus_wait_probe.py 1250 (compute-stream busy fraction)
#   gpurun_out/ev_<tag>/epoch_natural.json  tools/epoch_natural.py (tracker vs label-derived voicing on the bundled recordings)
#   gpurun_out/ev_<tag>/array_api.txt, mt_ladder.txt   tools/array_api_probe.py, tools/mt_ladder_probe.py
#   gpurun_out/ev_<tag>/fused_cr_check.txt  tools/fused_cr_check.py (constant-rate analysis: one kernel vs the staged pair)
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
E=$R/gpurun_out/ev_$TAG
mkdir -p $E
cd $R
bash tools/profile_round.sh $TAG > $E/profile_round.log 2>&1
timeout 2400 python -m pytest tests -m gpu -q > $E/gpu_tests.txt 2>&1
cp gpurun_out/tolerance_report.json $E/tolerance_report.json 2>/dev/null; tail -3 $E/gpu_tests.txt
BENCH_SHARE_DEVICE=1 BENCH_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --steps 20 --warmup 5 --no-e2e --traffic none \
    --no-cpu-baseline > $E/share_device_n8.log 2> $E/share_device_n8.err
grep '^{"metric"' $E/share_device_n8.log | tail -1 > $E/share_device_n8.json
timeout 600 python bench.py --workload corpus --utts 1250 > $E/corpus1250.log 2> $E/corpus1250.err
grep '^{' $E/corpus1250.log | tail -1 > $E/corpus1250.json
timeout 600 python tools/corpus_wait_probe.py 1250 $E/corpus_busy.json > $E/corpus_busy.txt 2>&1
timeout 600 python tools/epoch_natural.py $E/epoch_natural.json > $E/epoch_natural.txt 2>&1
timeout 300 python tools/array_api_probe.py > $E/array_api.txt 2>&1
timeout 300 python tools/mt_ladder_probe.py > $E/mt_ladder.txt 2>&1
timeout 600 python tools/fused_cr_check.py > $E/fused_cr_check.txt 2>&1
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $E/smoke.txt 2>&1
tail -1 $E/smoke.txt
