#!/usr/bin/env python3
"""Sum rocprofv3 --pmc counter_collection.csv rows per kernel.
usage: summarize_pmc.py <kernel-substring> <dir-with-pmc-passes>..."""
import collections
import csv
import glob
import sys

kern = sys.argv[1]
for d in sys.argv[2:]:
    for f in sorted(glob.glob(d + '/**/*_counter_collection.csv', recursive=True)):
        rows = [r for r in csv.DictReader(open(f)) if kern in r['Kernel_Name']]
        if not rows:
            continue
        agg = collections.defaultdict(float)
        disp = set()
        for r in rows:
            agg[r['Counter_Name']] += float(r['Counter_Value'])
            disp.add(r['Dispatch_Id'])
        r = rows[0]
        print(f"# {f}: {len(disp)} dispatch(es) of {r['Kernel_Name']}")
        print("# " + str({k: r[k] for k in ('VGPR_Count', 'Accum_VGPR_Count', 'SGPR_Count', 'LDS_Block_Size', 'Scratch_Size', 'Grid_Size', 'Workgroup_Size')}))
        for k, v in sorted(agg.items()):
            print(f"{k} {v:.6g}")
