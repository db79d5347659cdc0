# Per-(kernel, grid) launch times of the attention kernels of the offline bench, from a rocprofv3 kernel trace (GPU box).
# usage: bash tools/probes/run_attn_probe.sh <outdir under gpurun_out> [filter=k_attn]
# (Round 5 ran this once per value of an environment switch that removed parts of the kernels -- stores, the score phase, phase C --
#  and then with s_memrealtime stamps per phase; those hooks are not in the tree, the numbers are in DESIGN.md "Round 5".)
R=$PWD; O=$R/gpurun_out/$1; F=${2:-k_attn}; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace
rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-host-leg --no-secondary > $O/bench.json 2> $O/bench.err
echo "rc $?" > $O/summary.txt
python3 $R/tools/per_grid_stats.py $(ls $O/trace/*/*_kernel_trace.csv | head -1) 9 $F >> $O/summary.txt
rm -rf $O/trace
cat $O/summary.txt
