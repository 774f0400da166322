# round 5, GPU call 3: replicas + the router pipeline + `--gpus` (tests), then the k-mer loop's hash variants alone on the GPU
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/r05c; mkdir -p $out
timeout 600 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_cli.py -m gpu -x -q > $out/pytest_multi.txt 2>&1; tail -15 $out/pytest_multi.txt
./tools/hash_variants > $out/hash_variants.txt 2>&1; cat $out/hash_variants.txt
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE --kernel-include-regex 'loop_kernel' --output-format csv -d $out/pmc_hv -o s -- ./tools/hash_variants > /dev/null 2>&1
python tools/pmc_by_kernel.py $out/pmc_hv --last 4 > $out/hash_variants_pmc.json; rm -rf $out/pmc_hv
cat $out/hash_variants_pmc.json
