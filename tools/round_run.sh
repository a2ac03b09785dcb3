# Round evidence, one call on the GPU box:  bash tools/round_run.sh r04   ->  gpurun_out/<tag>/...
# (rocprofv3 is given the program itself after `--`; counters run in their own passes, never with trace domains)
set -e
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG
mkdir -p $O
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.txt 2>&1; tail -2 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
python bench.py --steps 30 --warmup 5 > $O/bench.json 2> $O/bench.err
echo bench done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --sustain-ms 0 > $O/bench_profiled.json 2> $O/bench_profiled.err
cp $(find $O/prof -name '*kernel_stats.csv' | head -1) $O/kernel_stats.csv
rm -rf $O/prof
echo kernel stats done
# N = 2 through bench.py's own launcher, both ranks on this box's one GPU (gloo): the N > 1 code path, not a measurement
python bench.py --gpus 2 --rehearse-on-one-gpu --steps 10 --warmup 2 --no-cpu-baseline --no-extras --sustain-ms 400 > $O/bench_rehearse2.json 2> $O/bench_rehearse2.err || echo "rehearsal failed"
python bench.py --grid 128 --batch 32 --points 120000 --steps 10 --warmup 2 --no-cpu-baseline --sustain-ms 1500 > $O/c3_bench.json 2> $O/c3.err
python3 tools/train_step_bench.py --graph --iters 20 > $O/train_bench.txt 2>&1 || true
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -- python3 tools/train_step_bench.py --graph --iters 20 > $O/train_profiled.txt 2>&1 || true
cp $(find $O/prof_train -name '*kernel_stats.csv' | head -1) $O/train_kernel_stats.csv || true
rm -rf $O/prof_train
python3 tools/train_step_bench.py --graph --iters 20 --bf16 > $O/train_bench_bf16.txt 2>&1 || true
python tools/c4_bench.py --batch 32 > $O/c4.txt 2>&1 || true
python tools/c4_bench.py --batch 32 --voxel-size 0.9 0.9 0.9 >> $O/c4.txt 2>&1 || true   # (0.9 m: the ~100 m scans fit 128^3)
python tools/conv_ab.py --rounds 3 > $O/conv_ab.txt 2>&1 || true
python tools/step_host_profile.py > $O/step_host.txt 2>&1 || true
python tools/k1_time.py > $O/k1_time.txt 2>&1 || true
python tools/corr_time.py > $O/corr_time.txt 2>&1 || true
# what the z-walk's hand-overs rest on, measured on the part (the protocol model and the ISA audit are CPU tests): an
# LDS-DMA's data against an LDS counter, DS order, M0 restored behind the load -- with a negative control
hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/micro/ldsdma_handover.hip -o /tmp/ldsdma_handover 2> $O/handover_build.log && timeout -k 5 120 /tmp/ldsdma_handover 400000 > $O/ldsdma_handover.txt 2>&1 || echo "handover probe failed"
# PMC passes
mkdir -p $O/pmc
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CYCLES -d $O/pmc/g1 --output-format csv -- python3 tools/profile_path.py --iters 3 --train > $O/pmc/g1.log 2>&1
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d $O/pmc/g2 --output-format csv -- python3 tools/profile_path.py --iters 3 --train > $O/pmc/g2.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $O/pmc/g3 --output-format csv -- python3 tools/profile_path.py --iters 3 --train > $O/pmc/g3.log 2>&1
python3 tools/pmc_summary.py $O/pmc/g1 $O/pmc/g2 $O/pmc/g3 > $O/pmc_summary.csv
rocprofv3 --pmc FETCH_SIZE -d $O/pmc/t1 --output-format csv -- python3 tools/profile_path.py --iters 3 > $O/pmc/t1.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $O/pmc/t2 --output-format csv -- python3 tools/profile_path.py --iters 3 > $O/pmc/t2.log 2>&1
rocprofv3 --pmc TCC_EA0_ATOMIC_sum -d $O/pmc/t3 --output-format csv -- python3 tools/profile_path.py --iters 3 > $O/pmc/t3.log 2>&1
python3 tools/pmc_summary.py $O/pmc/t1 $O/pmc/t2 $O/pmc/t3 > $O/pmc_traffic.csv
rm -rf $O/pmc
echo all done
