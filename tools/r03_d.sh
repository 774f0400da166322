# partition v3: GPU suite + sketch-stage kernel times under the tracer (three sketches of the bench read set)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r03d; mkdir -p $out
python -m pytest tests -m gpu -x -q > $out/pytest.txt 2>&1; grep -n "passed\|failed" $out/pytest.txt; grep -n "Error\|assert" $out/pytest.txt | head -20
rocprofv3 --kernel-trace --output-format csv -d $out/tr -o s -- python tools/trace_run.py > /dev/null 2> $out/trace_run.err
f=$(find $out/tr -name '*kernel_trace.csv' | head -1)
python tools/kernel_trace_summary.py $f --tail-ms 6 --top 16 | tee $out/sketch_kernels.md
rm -rf $out/tr; tail -4 $out/trace_run.err
