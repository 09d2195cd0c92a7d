import csv, collections, sys
want = sys.argv[1:] or ['render_fwd', 'render_bwd']
acc = collections.defaultdict(dict)
for d in 'abc':
    try:
        rows = list(csv.DictReader(open(f'/root/repo/gpurun_out/pmc_sq_{d}/p_counter_collection.csv')))
    except FileNotFoundError:
        continue
    tmp = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        n = r['Kernel_Name']
        for w in want:
            if w in n:
                tmp[w][r['Counter_Name']].append(float(r['Counter_Value']))
    for w, cs in tmp.items():
        for c, v in cs.items():
            acc[w][c] = sum(v) / len(v)
for w, cs in acc.items():
    print(w)
    wc = cs.get('SQ_WAVE_CYCLES', 1)
    for c in sorted(cs):
        print(f"   {c:24s} {cs[c]:14.0f}   {100*cs[c]/wc:6.1f}% of WAVE_CYCLES")
