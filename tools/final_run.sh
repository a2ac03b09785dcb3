set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -x -q > gpurun_out/final/pytest_gpu.txt 2>&1; tail -2 gpurun_out/final/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/final/smoke.txt 2>&1; tail -1 gpurun_out/final/smoke.txt
python bench.py --steps 30 --warmup 5 > gpurun_out/final/bench.json 2> gpurun_out/final/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/final/bench_profiled.json 2> gpurun_out/final/bench_profiled.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/final/prof_train -- python3 tools/train_step_bench.py --iters 20 > gpurun_out/final/train_profiled.txt 2>&1
python3 tools/train_step_bench.py --graph --iters 20 > gpurun_out/final/train_bench.txt 2>&1
python bench.py --grid 128 --batch 32 --points 120000 --steps 10 --warmup 2 --no-cpu-baseline > gpurun_out/final/c3_bench.json 2> gpurun_out/final/c3.err
python tools/dropin_forward_bench.py > gpurun_out/final/dropin.txt 2>&1 || true
python tools/c4_bench.py > gpurun_out/final/c4.txt 2>&1 || true
echo done
