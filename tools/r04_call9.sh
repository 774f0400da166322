cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r04l; mkdir -p $out
python tools/feed_bench.py 3333334 > $out/feed.txt 2> $out/feed.err
d=/tmp/feed_bench
export SYLPH_HIP_EXACT_DEDUP=1
for rep in 1 2; do
  echo "== one gz pair, run $rep"
  ( time SYLPH_HIP_FEED_TRACE=1 sylph_amd/sylph-hip sketch -1 $d/s_1.fq.gz -2 $d/s_2.fq.gz -d $d/out ) 2>&1 | grep -v "stretch" | grep -E "pgunzip|index of|timing|real|push|gather" | head -40
done
echo "== four gz pairs, -t 1"
( time SYLPH_HIP_FEED_TRACE=1 sylph_amd/sylph-hip sketch -1 $d/m0_1.fq.gz $d/m1_1.fq.gz $d/m2_1.fq.gz $d/m3_1.fq.gz -2 $d/m0_2.fq.gz $d/m1_2.fq.gz $d/m2_2.fq.gz $d/m3_2.fq.gz -d $d/out -t 1 ) 2>&1 | grep -v "stretch" | grep -E "pgunzip|index of|timing|real" | head -60
