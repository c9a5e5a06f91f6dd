# GPU box: SQ counters of the loss kernels (full-iteration bench), two separate --pmc passes
R=${GRAFT_REPO_ROOT:-$(pwd)}
rm -rf /tmp/lp; mkdir -p /tmp/lp $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --profile-steps 0 --loss l1_ssim --optimizer fused_adam --steps 4 --warmup 2"
timeout 150 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d /tmp/lp/pmc_sq -o l -- $B > /tmp/lp/a.log 2>&1
timeout 150 rocprofv3 --pmc SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS --output-format csv -d /tmp/lp/pmc_sq2 -o l -- $B > /tmp/lp/b.log 2>&1
timeout 150 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --output-format csv -d /tmp/lp/pmc_mem -o l -- $B > /tmp/lp/c.log 2>&1
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/lp/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'l1_ssim' in k or 'adam' in k:
            acc[k.split('(')[0][-40:]][r['Counter_Name']].append(float(r['Counter_Value']))
out = open('/root/repo/gpurun_out/r04_loss_kernel_counters.txt', 'w')
for k, d in acc.items():
    line = k + ': ' + ', '.join(f'{c}={sum(v)/len(v):.0f}' for c, v in sorted(d.items()))
    print(line); out.write(line + '\n')
PY
