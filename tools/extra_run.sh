set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04x
mkdir -p $O
# (a) C3's per-GPU share under the kernel trace
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --grid 128 --batch 32 --points 120000 --steps 10 --warmup 2 --no-cpu-baseline --no-extras --sustain-ms 0 > $O/c3_profiled.json 2> $O/c3_profiled.err
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/c3_kernel_stats.csv
rm -rf $O/prof
echo c3 done
# (b) the headline step at other batch sizes (tiles per launch): 8, 16, 64, 128
for B in 8 16 64 128; do
  python bench.py --batch $B --steps 20 --warmup 3 --no-cpu-baseline --no-extras --sustain-ms 0 > $O/bench_b$B.json 2> $O/bench_b$B.err || echo "batch $B failed"
done
echo batch done
# (c) the walk's lag (slots between a plane's fold and the first round that reads it) on the final build
for L in 2 3 4; do
  touch scene-net_amd/csrc/conv_i8s.hip && make -j16 EXTRA=-DSN_I8Z_LAG=$L > $O/build_lag$L.log 2>&1 && python tools/conv_ab.py --rounds 3 > $O/conv_ab_lag$L.txt 2>&1; grep "zwalk 1x2x12" $O/conv_ab_lag$L.txt
done
echo all done
