#!/bin/bash
# Re-create the rocprofv3 summaries under profiles/ (run on the GPU box from the repo root: bash tools/refresh_profiles.sh r01_v5).
# Batches in the offline run (--no-secondary: headline legs only): (2 warm-up + 8 timed) x 3 legs (HBM-resident, host memory pipelined, host memory
# synchronous GetResults) + 1 synchronous + 1 instrumented + 1 operator-level encoder pass for oracle_match.logits = 33;
# chunk steps in the streaming run: 64 warm-up + 64 timed + 1 instrumented = 129 (the divisors of tools/summarize_stats.py below).
# Every rocprofv3 run puts python3 directly after `--` and collects counters in their own passes (kernel-trace only).
set -e -o pipefail
TAG=${1:?tag, e.g. r01_v5}
R=$PWD
O=$R/gpurun_out/refresh_$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 8 --warmup 2 --no-secondary > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_write.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-secondary > /dev/null 2> $O/pmc_mfma.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_beam4 -- python3 $R/bench.py --beam 4 --steps 8 --warmup 2 --no-host-leg --no-cpu-baseline --no-secondary > /dev/null 2> $O/stats_beam4.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_conformer_zh -- python3 $R/bench.py --preset conformer-zh --batch 8 --seconds 30 --steps 8 --warmup 2 --no-host-leg --no-cpu-baseline --no-secondary > /dev/null 2> $O/stats_conformer_zh.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_streaming -- python3 $R/bench_streaming.py --streams 128 --seconds 20 --no-cpu-baseline > /dev/null 2> $O/stats_streaming.err
cd $R
python3 tools/summarize_stats.py $(ls $O/stats/*/*_kernel_stats.csv | head -1) 33 $O/${TAG}_offline_kernel_stats.csv > /dev/null
python3 tools/summarize_stats.py $(ls $O/stats_streaming/*/*_kernel_stats.csv | head -1) 129 $O/${TAG}_streaming_kernel_stats.csv > /dev/null
python3 tools/summarize_pmc.py $O/pmc_fetch FETCH_SIZE $O/${TAG}_pmc_fetch_summary.csv > /dev/null
python3 tools/summarize_pmc.py $O/pmc_write WRITE_SIZE $O/${TAG}_pmc_write_summary.csv > /dev/null
python3 tools/summarize_mfma.py $O/pmc_mfma $O/${TAG}_pmc_mfma_summary.csv $O/${TAG}_gemm_mfma_busy.json > /dev/null
python3 tools/make_gemm_traffic.py $O/${TAG}_pmc_fetch_summary.csv $O/${TAG}_pmc_write_summary.csv $O/${TAG}_gemm_traffic.json > /dev/null
python3 tools/make_hbm_kernels.py $O/${TAG}_pmc_fetch_summary.csv $O/${TAG}_pmc_write_summary.csv $O/${TAG}_offline_kernel_stats.csv $O/${TAG}_hbm_kernels.json > /dev/null
# (2 warm-up + 8 timed) + 1 synchronous + 1 instrumented = 12 batches in the beam-4 / conformer-zh runs
python3 tools/summarize_stats.py $(ls $O/stats_beam4/*/*_kernel_stats.csv | head -1) 12 $O/${TAG}_beam4_kernel_stats.csv > /dev/null
python3 tools/summarize_stats.py $(ls $O/stats_conformer_zh/*/*_kernel_stats.csv | head -1) 12 $O/${TAG}_conformer_zh_kernel_stats.csv > /dev/null
# The bench records come LAST and read the PMC summaries of THIS tag (K2HIP_PROFILE_DIR): what `roofline.traffic_note`,
# `roofline.committed_profile.source` and `roofline.hbm_kernels.from` cite are files of the set they are committed with
# (tests/test_abi.py::test_profiles_named_in_committed_bench_records_exist).  The record under rocprofv3 was taken before the summaries
# existed: its citations are rewritten from this run's line.
K2HIP_PROFILE_DIR=$O python3 bench.py --steps 20 --warmup 3 2> $O/bench.err | tail -1 > $O/${TAG}_bench.json   # carries the configs[2] / [3] / [4] legs under "secondary"
python3 - $O/bench_under_rocprof.json $O/${TAG}_bench.json $O/${TAG}_bench_under_rocprof.json <<'PY'
import json, sys
under = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
final = json.load(open(sys.argv[2]))
for k in ("traffic", "traffic_note", "committed_profile", "hbm_kernels"):
    if k in final.get("roofline", {}):
        under["roofline"][k] = final["roofline"][k]
json.dump(under, open(sys.argv[3], "w"))
PY
echo refreshed $TAG
