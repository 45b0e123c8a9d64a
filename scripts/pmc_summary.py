"""Summarise rocprofv3 --pmc csv output per kernel (mean counter value per dispatch)."""
import csv, glob, sys, collections, re
d = sys.argv[1]
f = glob.glob(d + '/**/*_counter_collection.csv', recursive=True)[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
meta = {}
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    m = re.search(r'(conv_\w+|concat_\w+|finalize_\w+|\w+_kernel)<?([^>(]*)', n)
    short = (m.group(1) + '<' + m.group(2) + '>') if m else n[:60]
    key = (short, r['Grid_Size'])
    agg[key][r['Counter_Name']].append(float(r['Counter_Value']))
    meta[key] = (r['LDS_Block_Size'], r['VGPR_Count'], r['SGPR_Count'])
for key, c in sorted(agg.items(), key=lambda kv: -sum(kv[1].get('SQ_BUSY_CYCLES', kv[1].get('SQ_WAIT_ANY', [0])))):
    if not any(s in key[0] for s in sys.argv[2:]) and len(sys.argv) > 2:
        continue
    print(key, 'LDS/VGPR/SGPR', meta[key])
    for name, v in sorted(c.items()):
        print('    %-28s %12.4g  (n=%d)' % (name, sum(v) / len(v), len(v)))
