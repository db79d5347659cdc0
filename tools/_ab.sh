R=$PWD
prof() {
  O=$R/gpurun_out/ab_$1; rm -rf $O; mkdir -p $O
  (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $O/s -- python3 $R/bench.py --preset conformer-zh --batch 8 --seconds 30 --steps 6 --warmup 2 --no-host-leg --no-cpu-baseline --no-secondary > $O/out.txt 2> $O/err.txt)
  grep -h "k_conformer_scores_softmax" $O/s/*/*_kernel_stats.csv | head -1 | sed -e "s/.*)\",//" | cut -c1-60
  tail -1 $O/out.txt | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('  step', d['ms_per_step'], 'enc', d['stages_ms_one_synchronous_batch']['encoder_ms'], d['results_sha1'])"
  rm -rf $O/s
}
echo strip16_nu; prof a
export K2HIP_CONFORMER_STRIP32=1
echo strip32; prof b
unset K2HIP_CONFORMER_STRIP32
echo strip16_nu_b; prof c
