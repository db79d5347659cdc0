# A/B of a streaming switch under rocprofv3 kernel-trace: per-kernel totals of the same run with and without it.
# usage (GPU box, repo root): bash tools/probes/run_stream_ab.sh <outdir under gpurun_out> <ENV_NAME>
set -e -o pipefail
R=$PWD; O=$R/gpurun_out/$1; SW=$2; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/a -- python3 $R/bench_streaming.py --seconds 6 --no-cpu-baseline > $O/bench_default.json 2> $O/err_a.txt
export $SW=1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b -- python3 $R/bench_streaming.py --seconds 6 --no-cpu-baseline > $O/bench_switch.json 2> $O/err_b.txt
cd $R
python3 tools/summarize_stats.py $(ls $O/a/*/*_kernel_stats.csv | head -1) 1 $O/stats_default.csv > /dev/null
python3 tools/summarize_stats.py $(ls $O/b/*/*_kernel_stats.csv | head -1) 1 $O/stats_switch.csv > /dev/null
python3 tools/per_grid_stats.py $(ls $O/a/*/*_kernel_trace.csv | head -1) 1 > $O/per_grid_default.txt
python3 tools/per_grid_stats.py $(ls $O/b/*/*_kernel_trace.csv | head -1) 1 > $O/per_grid_switch.txt
rm -rf $O/a $O/b
