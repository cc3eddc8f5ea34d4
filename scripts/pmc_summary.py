import csv, glob, collections, sys
base = sys.argv[1]
filt = sys.argv[2] if len(sys.argv) > 2 else ""
res = collections.OrderedDict()
for d in sorted(glob.glob(base + '/*/')):
    f = d + 'p_counter_collection.csv'
    try: rows = list(csv.DictReader(open(f)))
    except Exception as e: print(d, 'ERR', e); continue
    for r in rows:
        if filt and filt not in r['Kernel_Name']: continue
        key = (r['Kernel_Name'][:60], r['Grid_Size'])
        res.setdefault(key, collections.OrderedDict()).setdefault(r['Counter_Name'], []).append(float(r['Counter_Value']))
for (k, g), cs in res.items():
    if not any(x in k for x in ('gemm', 'tl_', 'attention', 'rows')): continue
    print(f"{k} grid={g}")
    for c, v in cs.items(): print(f"     {c:28s} mean={sum(v)/len(v):.5g} n={len(v)}")
