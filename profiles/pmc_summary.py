import csv, sys, collections, glob, re
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        m = re.search(r'(ft_\w+)(<[^>]*>)?', r['Kernel_Name'])
        if not m: continue
        k = m.group(1) + (m.group(2) or '')
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        cnt[(k, r['Counter_Name'])] += 1
    for k, d in agg.items():
        if 'rnn' not in k: continue
        print(k, 'launches', max(cnt[(k, c)] for c in d))
        for c, v in d.items():
            print(f'   {c:28s} per-launch {v / cnt[(k, c)]:14.1f}')
