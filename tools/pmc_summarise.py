"""Summarise a rocprofv3 --pmc run (counter_collection.csv [+ kernel_trace.csv]) per kernel: launches, mean counter
values per launch, mean duration.  One counter per rocprofv3 pass (FETCH_SIZE and WRITE_SIZE do not fit one pass on
gfx950): give every pass's output dir.  Usage: pmc_summarise.py <out.json> <rocprofv3 --output-format csv dir> [...]"""
import csv, glob, json, os, re, sys
from collections import defaultdict

dst, srcs = sys.argv[1], sys.argv[2:]


def short(name):
    m = re.search(r'(gemm_nt_mfma_kernel(?:_[bd])?(?:<(?:true|false)>)?|update_kernel|eval_ao_kernel\w*|Cijk_\w{0,24}|fft_rtc_\w+?_len\d+\w*?dim\d|pair_rows\w*|'
                  r'mul_coulG\w*|take_pivot\w*|z_r2c\w*kernel(?:<[^>]*>)?|z_c2r\w*kernel(?:<[^>]*>)?|strided_fft\w*kernel(?:<[^>]*>)?|gram_pivot_step_kernel|block_forward_kernel|skinny_rows_kernel|select_update\w*|'
                  r'square\w*kernel|transpose\w*kernel|trsm\w*|potrf\w*|larf\w*|syrk\w*)', name)
    return m.group(1) if m else name[:60]


res_all = {}
order = []
for src in srcs:
    cc = glob.glob(os.path.join(src, '**', '*counter_collection.csv'), recursive=True)
    kt = glob.glob(os.path.join(src, '**', '*kernel_trace.csv'), recursive=True)
    dur = {}
    for f in kt:
        for r in csv.DictReader(open(f)):
            dur[r.get('Dispatch_Id')] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-6
    per = defaultdict(lambda: defaultdict(float))
    names = {}
    for f in cc:
        for r in csv.DictReader(open(f)):
            d = r['Dispatch_Id']
            names[d] = short(r['Kernel_Name'])
            per[d][r['Counter_Name']] += float(r['Counter_Value'])
    out = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(int)
    for d, cs in per.items():
        k = names[d]
        cnt[k] += 1
        for c, v in cs.items():
            out[k][c] += v
        if d in dur:
            out[k]['ms'] += dur[d]
    for k in out:
        e = res_all.setdefault(k, {})
        e['launches'] = cnt[k]
        for c, v in out[k].items():
            e[('avg_ms' if c == 'ms' else c + '_per_launch')] = v / cnt[k]
res = dict(sorted(res_all.items(), key=lambda kv: -kv[1].get('avg_ms', 0) * kv[1]['launches']))
json.dump(res, open(dst, 'w'), indent=1)
for k, v in list(res.items())[:12]:
    print(k, v)
