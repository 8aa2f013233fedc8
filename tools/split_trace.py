import csv, sys
rows=list(csv.DictReader(open('gpurun_out/kstats/ks_kernel_trace.csv')))
for name in ('k_mu_cells','k_mu_lines','k_mu_finish'):
    sel=[r for r in rows if name in r['Kernel_Name']]
    d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in sel]
    if d: print(name, len(d), "first half avg %.2f  second half avg %.2f" % (sum(d[:len(d)//2])/(len(d)//2), sum(d[len(d)//2:])/(len(d)-len(d)//2)))
