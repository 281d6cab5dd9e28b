import csv, sys, collections, glob, re
pat = sys.argv[2]
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); dur = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        m = re.search(r'(ft_\w+)(<[^>]*>)?', r['Kernel_Name'])
        if not m: continue
        k = m.group(1) + (m.group(2) or '')
        if pat not in k: continue
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
        dur[k].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
    for k, d in agg.items():
        print(k, 'launches', max(cnt[(k, c)] for c in d), 'avg dur us %.1f' % (sum(dur[k]) / len(dur[k]) / 1e3))
        for c, v in sorted(d.items()):
            print(f'   {c:28s} per-launch {v / cnt[(k, c)]:16.1f}')
