# usage (GPU box, repo root): bash tools/probes/run_kernel_stats.sh <tag> [ENV=VALUE ...] -- kernel stats + per-grid durations of one
# rocprofv3 --kernel-trace pass over bench.py (8 + 2 steps); results under gpurun_out/ks_<tag>/
set -e -o pipefail
TAG=$1; shift
for kv in "$@"; do export "$kv"; done
R=$PWD; O=$R/gpurun_out/ks_$TAG; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -- python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/err.txt
cd $R
python3 tools/summarize_stats.py $(ls $O/s/*/*_kernel_stats.csv | head -1) 22 $O/stats.csv > /dev/null
python3 tools/per_grid_stats.py $(ls $O/s/*/*_kernel_trace.csv | head -1) 22 > $O/per_grid.txt
rm -rf $O/s/*/*_agent_info.csv
