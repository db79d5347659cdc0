set -e -o pipefail
R=$PWD; O=$R/gpurun_out/ks_conf; rm -rf $O; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  if [ $v = 1 ]; then export K2HIP_CONFORMER_SCATTER_V1=1; else unset K2HIP_CONFORMER_SCATTER_V1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/s$v -- python3 $R/bench.py --preset conformer-zh --batch 8 --seconds 30 --steps 6 --warmup 2 --no-cpu-baseline --no-secondary --no-host-leg > $O/bench$v.json 2> $O/err$v.txt
  python3 $R/tools/summarize_stats.py $(ls $O/s$v/*/*_kernel_stats.csv | head -1) 8 $O/stats$v.csv > /dev/null
  rm -rf $O/s$v
done
