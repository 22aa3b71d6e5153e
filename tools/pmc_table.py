"""Joins rocprofv3 --pmc counter_collection.csv files into one row per dispatch (by dispatch order)."""
import csv, sys, re, collections
rows = collections.OrderedDict()
for p in sys.argv[1:]:
    for r in csv.DictReader(open(p)):
        k = int(r['Dispatch_Id'])
        d = rows.setdefault(k, {'kernel': re.search(r'(k_\w+(<[^>]*>)?)', r['Kernel_Name']).group(1) if 'k_' in r['Kernel_Name'] else r['Kernel_Name'][:30],
                                'grid': int(r['Grid_Size']), 'wg': int(r['Workgroup_Size']), 'vgpr': r['VGPR_Count'], 'agpr': r['Accum_VGPR_Count'],
                                'dur_us': (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3})
        d[r['Counter_Name']] = float(r['Counter_Value'])
names = []
for d in rows.values():
    for k in d:
        if k not in names: names.append(k)
print('\t'.join(['id'] + names))
for k, d in rows.items():
    print('\t'.join([str(k)] + [('%.0f' % d[n] if isinstance(d.get(n), float) and n != 'dur_us' else ('%.1f' % d[n] if n == 'dur_us' else str(d.get(n, '')))) for n in names]))
