# PMC passes for the two largest kernels after the dominant one (bench.py C3 as the workload; counters only)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
out=gpurun_out/final2; mkdir -p $out
for c in FETCH_SIZE WRITE_SIZE; do
rocprofv3 --pmc $c --kernel-include-regex 'probe_kernel|bucket_replay_kernel<256' --output-format csv -d $out/$c -o s -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-include-regex 'probe_kernel|bucket_replay_kernel<256' --output-format csv -d $out/SQ -o s -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python - <<'PY'
import csv, glob, collections
for d in ('FETCH_SIZE','WRITE_SIZE','SQ'):
    for f in glob.glob(f'gpurun_out/final2/{d}/*counter_collection.csv'):
        agg=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k='probe' if 'probe_kernel' in r['Kernel_Name'] else 'replay'
            agg[(k, r['Counter_Name'])].append(float(r['Counter_Value']))
        for k,v in sorted(agg.items()):
            v=v[-3:]
            print(k[0], k[1], '%.4g' % (sum(v)/len(v)))
PY
